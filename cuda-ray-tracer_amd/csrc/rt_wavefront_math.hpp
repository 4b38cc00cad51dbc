// rt_wavefront_math.hpp -- the per-lane arithmetic of the wavefront kernel (rt_wavefront.hip) that is not already in
// rt_math.hpp: per-class polynomial coefficients, the "does the reference's solver produce a root" predicates, the ray
// monomials in three parts, and the conservative culling tests (primary cone, tile pyramid, shadow-phase records).
//
// It lives in a header of its own so that tools/count_flops.cpp can run exactly this code -- the very functions the kernel is
// compiled from -- over an operation-counting scalar (the flop accounting of bench.py).  Nothing here uses a
// cross-lane or memory builtin; the kernel wraps these predicates in ballots and loops.
#pragma once

#include "rt_math.hpp"

namespace rtm {

// Does the reference's solver compute a root for these coefficients (include/surface_impl.h:138-154)?
// false: it returns -1 without a division (negative discriminant, or a constant polynomial).
__device__ __forceinline__ bool needs_solve(double t2, double t1, double t0)
{
    if (fabs(t2) > EPS) {
        double delta = t1 * t1 - 4.0 * t2 * t0;
        return !(delta < 0);
    }
    return fabs(t1) > EPS;
}

// Unit spheres (t2 = |d|^2 > 0): does the reference's solver produce a root that can pass a "t >= EPS" test?
//   * discriminant < 0                      -> it returns -1                      (include/surface_impl.h:141-144)
//   * t1 > 0 and t0 > 0 (origin outside, moving away) -> both roots are <= 0: with t2, t0 > 0 the computed
//     discriminant is <= fl(t1*t1), a correctly rounded sqrt of that is <= t1, so (-t1 + sqrt)/(2 t2) <= 0 and
//     (-t1 - sqrt)/(2 t2) < 0 in the reference's own floating-point evaluation, not just in exact arithmetic.
// Either way neither the nearest-hit test (t >= EPS) nor the shadow test (t > EPS) can accept, so the sqrt and the
// divisions need not be executed.  This removes the "own sphere" solve of every shadow ray that leaves a lit surface.
__device__ __forceinline__ bool us_needs_solve(bool quad, double four_t2, double t1, double t0)
{
    if (quad) { // (no short-circuit: three compares and two mask operations instead of a divergent branch inside the callers' loops)
        const bool no_root = t1 * t1 - four_t2 * t0 < 0;
        const bool away = (t1 > 0.0) & (t0 > 0.0);
        return !(no_root | away);
    }
    return fabs(t1) > EPS;
}

// t1 / t0 per class table entry (rt_scene_dev.h).  Same sums as rtm::quadric_poly / include/surface_impl.h:54-103
// with the exactly-zero groups left out (see RT_CLS_* in rt_scene_dev.h for why that is exact).  t1 and t0 are
// separate functions because primary rays take t0 from a per-object table (it depends on the origin only).
__device__ __forceinline__ double us_t1(const UsEntry &e, const Mono &m)
{
    return ((m.u1 + e.kx * m.d.x) + e.ky * m.d.y) + e.kz * m.d.z;
}
__device__ __forceinline__ double us_t0(const UsEntry &e, const Mono &m)
{
    return (((m.u0 + e.kx * m.o.x) + e.ky * m.o.y) + e.kz * m.o.z) + e.c;
}
__device__ __forceinline__ double lin_t1(const LinEntry &e, const Mono &m)
{
    return (e.kx * m.d.x + e.ky * m.d.y) + e.kz * m.d.z;
}
__device__ __forceinline__ double lin_t0(const LinEntry &e, const Mono &m)
{
    return ((e.kx * m.o.x + e.ky * m.o.y) + e.kz * m.o.z) + e.c;
}
__device__ __forceinline__ double gq_t2(const GqEntry &e, const Mono &m)
{
    return ((((e.x2 * m.dxx + e.y2 * m.dyy) + e.z2 * m.dzz) + e.xy * m.dxy) + e.xz * m.dxz) + e.yz * m.dyz;
}
__device__ __forceinline__ double gq_t1(const GqEntry &e, const Mono &m)
{
    return (((((((e.x2 * m.sx + e.y2 * m.sy) + e.z2 * m.sz) + e.xy * m.cxy) + e.xz * m.cxz) + e.yz * m.cyz) + e.kx * m.d.x) +
            e.ky * m.d.y) + e.kz * m.d.z;
}
__device__ __forceinline__ double gq_t0(const GqEntry &e, const Mono &m)
{
    return ((((((((e.x2 * m.oxx + e.y2 * m.oyy) + e.z2 * m.ozz) + e.xy * m.oxy) + e.xz * m.oxz) + e.yz * m.oyz) + e.kx * m.o.x) +
             e.ky * m.o.y) + e.kz * m.o.z) + e.c;
}

// Monomials of a ray in three parts, so that a part that is shared (origin of a whole chunk, direction of a
// directional light) is formed once.  The cross / mixed ones are only formed when some table needs them.
template <bool NEED_CROSS>
__device__ __forceinline__ void mono_set_o(Mono &m, const D3 &o)
{
    constexpr bool need_cross = NEED_CROSS;
    m.o = o;
    m.oxx = o.x * o.x;
    m.oyy = o.y * o.y;
    m.ozz = o.z * o.z;
    m.u0 = (m.oxx + m.oyy) + m.ozz;
    m.oxy = m.oxz = m.oyz = 0.0;
    if (need_cross) {
        m.oxy = o.x * o.y;
        m.oxz = o.x * o.z;
        m.oyz = o.y * o.z;
    }
}
template <bool NEED_CROSS>
__device__ __forceinline__ void mono_set_d(Mono &m, const D3 &d)
{
    constexpr bool need_cross = NEED_CROSS;
    m.d = d;
    m.dxx = d.x * d.x;
    m.dyy = d.y * d.y;
    m.dzz = d.z * d.z;
    m.u2 = (m.dxx + m.dyy) + m.dzz;
    m.dxy = m.dxz = m.dyz = 0.0;
    if (need_cross) {
        m.dxy = d.x * d.y;
        m.dxz = d.x * d.z;
        m.dyz = d.y * d.z;
    }
}
template <bool NEED_CROSS>
__device__ __forceinline__ void mono_set_od(Mono &m)
{
    constexpr bool need_cross = NEED_CROSS;
    m.sx = 2.0 * m.o.x * m.d.x;
    m.sy = 2.0 * m.o.y * m.d.y;
    m.sz = 2.0 * m.o.z * m.d.z;
    m.u1 = (m.sx + m.sy) + m.sz;
    m.cxy = m.cxz = m.cyz = 0.0;
    if (need_cross) {
        m.cxy = m.o.x * m.d.y + m.d.x * m.o.y;
        m.cxz = m.o.x * m.d.z + m.d.x * m.o.z;
        m.cyz = m.o.y * m.d.z + m.d.y * m.o.z;
    }
}

// normal_vector (include/surface_impl.h:157-172) of a unit sphere from its table entry.  With x2 = y2 = z2 = 1 and every other
// coefficient of degree >= 2 exactly zero, each term of the reference's gradient that carries such a coefficient is an exact
// signed zero, and adding it changes nothing but possibly the sign of a zero component (which no later operation can turn into
// a different pixel: it only ever multiplies or is added to something); what remains, in the reference's order, is
// ((2 * 1) * p + k), then the same normalisation.
__device__ __forceinline__ D3 sphere_normal(const UsEntry &e, const D3 &p)
{
    const D3 g{(2.0 * 1.0) * p.x + e.kx, (2.0 * 1.0) * p.y + e.ky, (2.0 * 1.0) * p.z + e.kz};
    return normalize3(g);
}

// Nearest-hit rule of src/update-cpu.cpp:52-55 made order-independent: strict '<' with ascending object index
// means the lowest index wins ties.
__device__ __forceinline__ void accept(double t, int k, double &best_t, int &best)
{
    if (t >= EPS && t < MAX_T && (t < best_t || (t == best_t && k < best))) {
        best_t = t;
        best = k;
    }
}

// Conservative culling for primary rays: which unit spheres can ANY of this wave's 64 primary rays hit?
// All rays leave the camera origin; they lie in the cone of half-angle theta around `axis` (the direction of
// one central lane), theta = the largest angle between axis and a lane's direction.  A sphere (centre v
// relative to the origin, radius r) can only be hit if it reaches into that cone; with h = v.axis and
// rho = distance of the centre from the axis line, rho cos(theta) - h sin(theta) is the signed distance of the
// centre from the cone's generator line (never larger than its distance to the cone), so the sphere is skipped
// only when that exceeds r plus a margin (1e-6 relative + the cancellation error of the reference's own t0 for
// huge coordinates).  Squared form, no sqrt / division.
__device__ __forceinline__ bool sphere_in_cone(double kx, double ky, double kz, double r, double inv_r, const D3 &org, const D3 &axis,
                                               double cos_t)
{
    bool rel;
    {
        // The squared comparison below (and the corner-pixel bound on the half-angle) needs a cone narrower than a
        // half-space.  A 16-pixel block only gets that wide with absurd aspect ratios (a 106 x 2 image at 86 degrees:
        // found by tests/tools/fuzz_parity.py), but then nothing is culled.
        if (!(r < INFINITY) || !(cos_t > 0.2)) {
            rel = true;
        } else {
            const double ccx = -0.5 * kx, ccy = -0.5 * ky, ccz = -0.5 * kz;
            const double vx = ccx - org.x, vy = ccy - org.y, vz = ccz - org.z;
            const double vv = vx * vx + vy * vy + vz * vz;
            const double h = vx * axis.x + vy * axis.y + vz * axis.z;
            double rho2 = vv - h * h;
            rho2 = rho2 > 0.0 ? rho2 : 0.0;
            const double v1 = fabs(vx) + fabs(vy) + fabs(vz);
            const double s2 = ccx * ccx + ccy * ccy + ccz * ccz + org.x * org.x + org.y * org.y + org.z * org.z;
            const double c = cos_t * (1.0 - 1e-9);               // a slightly wider cone
            double sin2 = 1.0 - c * c;
            sin2 = sin2 > 0.0 ? sin2 : 0.0;
            const double sn = sqrt(sin2);
            const double lim = r + 1e-6 * (v1 + r + 1.0) + 1e-12 * (s2 + 1.0) * inv_r;
            const double rhs = lim + h * sn; // need rho * c <= rhs
            rel = !(rhs < 0.0) && !(rho2 * c * c > rhs * rhs);
        }
    }
    return rel;
}

// Tile-level early-out for all-sphere scenes.  The rays of a tile are t * M3 * (cx, cy, 1), t > 0, with cx / cy between
// the camera-plane coordinates of the tile's first and last pixel (widened by half a pixel), i.e. they lie inside the
// pyramid of five planes through the ray origin whose normals are M3^-T (1, 0, -cx0), (-1, 0, cx1), (0, 1, -cy0),
// (0, -1, cy1), (0, 0, 1) (FrameArgs::tile_nt; no normalisation, no division, no square root).  A sphere whose centre lies
// further than its radius plus the margin of sphere_in_cone outside ANY of the planes cannot be hit by a ray of the tile;
// the comparison is made on squares.  Purely conservative: the verdict only decides whether phase A runs at all.
struct TilePlanes {
    D3 n[5];
    double nn[5]; // n . n
};

__device__ __forceinline__ TilePlanes tile_planes(const FrameArgs &fa, double cx0, double cx1, double cy0, double cy1)
{
    const D3 c0{fa.tile_nt[0], fa.tile_nt[1], fa.tile_nt[2]}, c1{fa.tile_nt[3], fa.tile_nt[4], fa.tile_nt[5]}, c2{fa.tile_nt[6], fa.tile_nt[7], fa.tile_nt[8]};
    TilePlanes P;
    P.n[0] = D3{c0.x - cx0 * c2.x, c0.y - cx0 * c2.y, c0.z - cx0 * c2.z};
    P.n[1] = D3{cx1 * c2.x - c0.x, cx1 * c2.y - c0.y, cx1 * c2.z - c0.z};
    P.n[2] = D3{c1.x - cy0 * c2.x, c1.y - cy0 * c2.y, c1.z - cy0 * c2.z};
    P.n[3] = D3{cy1 * c2.x - c1.x, cy1 * c2.y - c1.y, cy1 * c2.z - c1.z};
    P.n[4] = c2;
#pragma unroll
    for (int k = 0; k < 5; k++) P.nn[k] = dot3(P.n[k], P.n[k]);
    return P;
}

__device__ __forceinline__ bool sphere_in_pyramid(double kx, double ky, double kz, double r, double inv_r, const D3 &org, const TilePlanes &P)
{
    if (!(r < INFINITY)) return true;
    const double ccx = -0.5 * kx, ccy = -0.5 * ky, ccz = -0.5 * kz;
    const D3 v{ccx - org.x, ccy - org.y, ccz - org.z};
    const double v1 = fabs(v.x) + fabs(v.y) + fabs(v.z);
    const double s2 = ccx * ccx + ccy * ccy + ccz * ccz + org.x * org.x + org.y * org.y + org.z * org.z;
    const double lim = r + 1e-6 * (v1 + r + 1.0) + 1e-12 * (s2 + 1.0) * inv_r; // as in sphere_in_cone
    const double lim2 = lim * lim * (1.0 + 1e-9);
    bool in = true;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const double f = dot3(P.n[k], v);
        in = in && !(f < 0.0 && f * f > lim2 * P.nn[k]); // beyond plane k by more than lim: outside
    }
    return in;
}

// Conservative culling for phase B: can unit sphere (base + lane) block ANY shadow ray of this chunk towards
// this light?  Returns the wave-uniform mask of table entries that must be tested.
//
// Every shadow ray of the chunk starts within `ball.R` of `ball.c` (hit point + 1e-2 * unit normal; R already
// includes that bias) and runs along sdir (directional light: the same FP32-rounded direction for all) or to
// within 1e-2 + 6e-8|e| of the light position (point light, parameter range (EPS, 1)).  A sphere can only block
// if the reference's solver finds a root, i.e. if the ray's line (directional) / segment (point light) comes
// within the sphere's radius of its centre.  By the triangle inequality that requires the centre to be within
// r + R of the chunk's axis line / segment.  `lim` pads this with a margin that dwarfs every rounding
// effect involved (1e-6 relative to the distances, plus the cancellation error of the reference's own t0 when
// coordinates are huge); a larger margin only means a few more objects get tested.  Spheres without a real
// radius carry r = +inf and are always tested; other classes are never culled.  No division, no sqrt.
struct Ball {
    double cx, cy, cz, R;
};

template <bool SPHERICAL, typename Light> // the light's kind, known to the caller (wave-uniform): only that half is compiled into the loop.  Light: a DevLight in
                                           // LDS, or in the constant address space (the lean path reads the lights with scalar loads)
__device__ __forceinline__ bool sphere_relevant(const UsEntry &e, const Ball &ball, const Light &lt)
{
    bool rel = false;
    const double r = e.r;
    if (!(r < INFINITY)) {
        rel = true; // +inf (not cullable) or NaN
    } else {
        const double ccx = -0.5 * e.kx, ccy = -0.5 * e.ky, ccz = -0.5 * e.kz; // centre (src/surface.cpp:10-12 inverted)
        const double wx = ccx - ball.cx, wy = ccy - ball.cy, wz = ccz - ball.cz;
        const double ww = wx * wx + wy * wy + wz * wz;
        const double w1 = fabs(wx) + fabs(wy) + fabs(wz); // >= |w|
        const double s2 = ccx * ccx + ccy * ccy + ccz * ccz + ball.cx * ball.cx + ball.cy * ball.cy + ball.cz * ball.cz;
        if (!SPHERICAL) {
            const double along = wx * lt.sdir[0] + wy * lt.sdir[1] + wz * lt.sdir[2];
            const double perp2 = ww - along * along * lt.inv_uu;
            const double lim = r + ball.R + 1e-6 * (w1 + r + ball.R + 1.0) + 1e-12 * (s2 + 1.0) * e.inv_r;
            // within reach of the axis, and not entirely behind the chunk (roots must be > EPS)
            rel = !(perp2 > lim * lim) && !(along < -lim * lt.len_u);
        } else {
            const double ex = lt.p[0] - ball.cx, ey = lt.p[1] - ball.cy, ez = lt.p[2] - ball.cz;
            const double ee = ex * ex + ey * ey + ez * ez;
            const double e1 = fabs(ex) + fabs(ey) + fabs(ez);
            const double we = wx * ex + wy * ey + wz * ez;
            const double l2 = lt.p[0] * lt.p[0] + lt.p[1] * lt.p[1] + lt.p[2] * lt.p[2];
            const double lim = r + ball.R + 1e-6 * (w1 + r + ball.R + e1 + 1.0) + 1e-12 * (s2 + l2 + 1.0) * e.inv_r;
            const double lim2 = lim * lim;
            // squared distance of the centre from the segment [ball.c, light]: closest point at parameter
            // we/ee clamped to [0, 1]; the middle case is compared multiplied through by ee
            if (!(we > 0.0)) rel = !(ww > lim2);
            else if (!(we < ee)) rel = !((ww - 2.0 * we) + ee > lim2);
            else rel = !(ww * ee - we * we > lim2 * ee);
        }
    }
    return rel;
}

// The light-independent half of relevant_mask for one (chunk, sphere): formed once per chunk in phase A' (lane = sphere) and
// kept in LDS, so that a directional light's culling decision is eight operations per sphere instead of fifty.  Same
// operations in the same order as relevant_mask, hence the same decisions.  Spheres that are never culled carry lim = +inf
// (NaN radii give NaN: every comparison below is then false, i.e. "test it").
struct alignas(16) CullRec {
    double wx, wy, wz, ww, lim, limr; // limr = lim without the ball's radius: the sphere's own reach, for the box test below
};                            // 48 B: three 16-byte LDS reads, conflict-free at this stride
// Half extents of the chunk's bounding box (same centre as its ball), the 1e-2 shadow bias of the ray origins included.
struct alignas(16) BoxH {
    double hx, hy, hz, pad;
};

__device__ __forceinline__ CullRec cull_record(const UsEntry &e, const Ball &ball)
{
    CullRec c;
    const double r = e.r;
    const double ccx = -0.5 * e.kx, ccy = -0.5 * e.ky, ccz = -0.5 * e.kz;
    c.wx = ccx - ball.cx; c.wy = ccy - ball.cy; c.wz = ccz - ball.cz;
    c.ww = c.wx * c.wx + c.wy * c.wy + c.wz * c.wz;
    const double w1 = fabs(c.wx) + fabs(c.wy) + fabs(c.wz);
    const double s2 = ccx * ccx + ccy * ccy + ccz * ccz + ball.cx * ball.cx + ball.cy * ball.cy + ball.cz * ball.cz;
    const double margin = 1e-6 * (w1 + r + ball.R + 1.0) + 1e-12 * (s2 + 1.0) * e.inv_r;
    c.lim = r + ball.R + margin;
    c.limr = r + margin;
    if (!(r < INFINITY)) c.lim = c.limr = r; // +inf or NaN: always tested
    return c;
}

__device__ __forceinline__ bool crec_relevant(const CullRec &c, const D3 &sdir, double inv_uu, double len_u)
{
    const double along = c.wx * sdir.x + c.wy * sdir.y + c.wz * sdir.z;
    const double perp2 = c.ww - along * along * inv_uu;
    return !(perp2 > c.lim * c.lim) && !(along < -c.lim * len_u);
}

// Second stage for a directional light, used when the ball lets many spheres through (a chunk whose hits lie on a near and a far
// object has a long thin box and a fat ball).  All shadow rays of the chunk start inside the box and run along sdir, so in the plane
// perpendicular to sdir they lie inside the box's shadow, a hexagon whose edge normals are sdir x e_k; a sphere can only be met if
// its centre's projection is within its reach of that hexagon, hence within reach of it along each of the three normals:
//     |w . (sdir x e_k)|  <=  sum_j h_j |e_j . (sdir x e_k)|  +  reach * |sdir x e_k|,        w = centre - box centre.
// |sdir x e_k| is bounded by the 1-norm of its two components (no square root; the bound only has to be conservative).
// reach = limr carries the same margins as the ball test (1e-6 of the distances involved), which dwarf the rounding of the
// two-term products here.  NaN / inf reach: every comparison is false, the sphere stays.
// (s_yz = |sdir.y| + |sdir.z| etc.: per-light constants, formed by the caller or taken from the light table)
__device__ __forceinline__ bool crec_in_box_shadow(const CullRec &c, const BoxH &h, const D3 &sdir, double s_yz, double s_xz, double s_xy)
{
    const double ax = fabs(sdir.x), ay = fabs(sdir.y), az = fabs(sdir.z);
    const double px = c.wy * sdir.z - c.wz * sdir.y, bx = (h.hy * az + h.hz * ay) + c.limr * s_yz;
    const double py = c.wz * sdir.x - c.wx * sdir.z, by = (h.hz * ax + h.hx * az) + c.limr * s_xz;
    const double pz = c.wx * sdir.y - c.wy * sdir.x, bz = (h.hx * ay + h.hy * ax) + c.limr * s_xy;
    return !(fabs(px) > bx * (1.0 + 1e-9)) && !(fabs(py) > by * (1.0 + 1e-9)) && !(fabs(pz) > bz * (1.0 + 1e-9));
}
__device__ __forceinline__ bool crec_in_box_shadow(const CullRec &c, const BoxH &h, const D3 &sdir)
{
    const double ax = fabs(sdir.x), ay = fabs(sdir.y), az = fabs(sdir.z);
    return crec_in_box_shadow(c, h, sdir, ay + az, ax + az, ax + ay);
}

__device__ __forceinline__ void blend(F3 &res, float ratio, const F3 &c)
{
    // UPDATE_COLOR, src/update-cpu.cpp:100
    res.x = (1.0f - ratio) * res.x + ratio * c.x;
    res.y = (1.0f - ratio) * res.y + ratio * c.y;
    res.z = (1.0f - ratio) * res.z + ratio * c.z;
}

__host__ __device__ inline uint32_t align16(uint32_t v) { return (v + 15u) & ~15u; }

} // namespace rtm
