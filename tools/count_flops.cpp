// count_flops.cpp -- FP64 operations per unit of work of the wavefront kernel, by the counting-scalar technique SURVEY.md 8(d)
// prescribes: the kernel's own math headers (rt_math.hpp, rt_wavefront_math.hpp -- the code rt_wavefront.hip is compiled
// from) are compiled for the host with `double` replaced by a scalar that counts add / sub / mul / div / sqrt as one
// operation each (the strict build has no FMA contraction; comparisons, negation, fabs and conversions are free; cbrt /
// acos / cos are counted separately), and every unit below is executed once on representative operands.
//
//   g++ -std=c++17 -O1 -Itools/flopcount_shim -Icuda-ray-tracer_amd/csrc tools/count_flops.cpp -o tools/bin/count_flops
//   tools/bin/count_flops > profiles/flop_table.json
//
// bench.py multiplies these costs with the device's work counters (rt_get_counters_detail).
#include <cmath>
#include <cstddef> // (before `double` is redefined below: rt_scene_dev.h includes it too)
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <type_traits>

static long g_ops = 0, g_special = 0, g_div = 0, g_sqrt = 0;

struct Counted {
    double v;
    constexpr Counted() : v(0.0) {}
    template <typename T, typename = std::enable_if_t<std::is_arithmetic<T>::value>>
    constexpr Counted(T x) : v((double) x) {}
    explicit operator float() const { return (float) v; }
    explicit operator int() const { return (int) v; }
    explicit operator bool() const { return v != 0.0; }
    Counted &operator+=(const Counted &o) { g_ops++; v += o.v; return *this; }
    Counted &operator-=(const Counted &o) { g_ops++; v -= o.v; return *this; }
    Counted &operator*=(const Counted &o) { g_ops++; v *= o.v; return *this; }
    Counted &operator/=(const Counted &o) { g_ops++; g_div++; v /= o.v; return *this; }
    friend Counted operator+(const Counted &a, const Counted &b) { g_ops++; return Counted(a.v + b.v); }
    friend Counted operator-(const Counted &a, const Counted &b) { g_ops++; return Counted(a.v - b.v); }
    friend Counted operator*(const Counted &a, const Counted &b) { g_ops++; return Counted(a.v * b.v); }
    friend Counted operator/(const Counted &a, const Counted &b) { g_ops++; g_div++; return Counted(a.v / b.v); }
    friend Counted operator-(const Counted &a) { return Counted(-a.v); }
    friend bool operator<(const Counted &a, const Counted &b) { return a.v < b.v; }
    friend bool operator>(const Counted &a, const Counted &b) { return a.v > b.v; }
    friend bool operator<=(const Counted &a, const Counted &b) { return a.v <= b.v; }
    friend bool operator>=(const Counted &a, const Counted &b) { return a.v >= b.v; }
    friend bool operator==(const Counted &a, const Counted &b) { return a.v == b.v; }
    friend bool operator!=(const Counted &a, const Counted &b) { return a.v != b.v; }
};
static_assert(sizeof(Counted) == 8, "layout of the scene records must not change");
inline Counted sqrt(const Counted &a) { g_ops++; g_sqrt++; return Counted(std::sqrt(a.v)); }
inline Counted fabs(const Counted &a) { return Counted(std::fabs(a.v)); }
inline Counted fmax(const Counted &a, const Counted &b) { return Counted(std::fmax(a.v, b.v)); }
inline Counted copysign(const Counted &a, const Counted &b) { return Counted(std::copysign(a.v, b.v)); }
inline Counted fma(const Counted &a, const Counted &b, const Counted &c) { g_ops += 2; return Counted(std::fma(a.v, b.v, c.v)); } // a multiplication and an addition
inline Counted cbrt(const Counted &a) { g_special++; return Counted(std::cbrt(a.v)); }
inline Counted acos(const Counted &a) { g_special++; return Counted(std::acos(a.v)); }
inline Counted cos(const Counted &a) { g_special++; return Counted(std::cos(a.v)); }

#define double Counted
#include "rt_wavefront_math.hpp"
#undef double

using namespace rtm;

static long g_last_div = 0, g_last_sqrt = 0;
template <typename F>
static long ops(F f)
{
    const long before = g_ops, d0 = g_div, s0 = g_sqrt;
    f();
    g_last_div = g_div - d0;
    g_last_sqrt = g_sqrt - s0;
    return g_ops - before;
}
struct DS { double div, sqrt; };
static DS last() { return DS{(double) g_last_div, (double) g_last_sqrt}; }

static volatile bool g_sink;

int main()
{
    // representative operands: a ray from the origin towards +z, a unit sphere table entry in front of it, a general quadric,
    // a plane, the clebsch cubic (all values generic: no term of the code below is skipped because of a zero)
    const D3 o{0.3, -0.2, 0.1}, d{0.12, -0.07, 0.99};
    UsEntry us{};
    us.kx = -2.0 * 1.5; us.ky = -2.0 * 0.5; us.kz = -2.0 * 9.0; us.c = 1.5 * 1.5 + 0.5 * 0.5 + 81.0 - 4.0; us.r = 2.0; us.inv_r = 0.5;
    GqEntry gq{};
    gq.x2 = 1.0; gq.y2 = 4.0; gq.z2 = 0.5; gq.xy = 0.5; gq.xz = -0.3; gq.yz = 0.2; gq.kx = 0.1; gq.ky = -0.2; gq.kz = -6.0; gq.c = -9.0;
    LinEntry lin{};
    lin.kx = 0.05; lin.ky = 1.0; lin.kz = 0.02; lin.c = 7.0;
    Counted cub[20];
    for (int i = 0; i < 20; i++) cub[i] = 0.1 * (i + 1) * ((i % 3) ? 1.0 : -1.0);
    DevLight dir_light{}, pt_light{};
    dir_light.p[0] = 0.3; dir_light.p[1] = 0.9; dir_light.p[2] = -0.3;
    for (int k = 0; k < 3; k++) dir_light.sdir[k] = (Counted) (float) dir_light.p[k];
    dir_light.inv_uu = 1.0; dir_light.len_u = 1.001; dir_light.u2 = 1.0;
    pt_light.spherical = 1; pt_light.p[0] = 4.0; pt_light.p[1] = 8.0; pt_light.p[2] = 3.0;
    FrameArgs fa{};
    for (int k = 0; k < 16; k++) fa.cam[k] = (k % 5 == 0) ? 1.0 : 0.01 * k;
    fa.origin[0] = 0.1; fa.origin[1] = 0.2; fa.origin[2] = 0.3;
    for (int k = 0; k < 9; k++) fa.tile_nt[k] = (k % 4 == 0) ? 1.0 : 0.02 * k;
    fa.cx_a = 0.001; fa.cx_b = -0.9; fa.cy_a = 0.001; fa.cy_b = -0.5;

    Mono m;
    const long mono_o = ops([&] { mono_set_o<false>(m, o); }), mono_d = ops([&] { mono_set_d<false>(m, d); }), mono_od = ops([&] { mono_set_od<false>(m); });
    Mono mc;
    const long mono_o_x = ops([&] { mono_set_o<true>(mc, o); }), mono_d_x = ops([&] { mono_set_d<true>(mc, d); }), mono_od_x = ops([&] { mono_set_od<true>(mc); });
    const Counted four_t2 = 4.0 * m.u2;

    // --- one executed test per surface class: coefficients + "does the solver produce a root" (sqrt / divisions are deferred: solve_*)
    const long test_unitsq = ops([&] { Counted t1 = us_t1(us, m), t0 = us_t0(us, m); g_sink = us_needs_solve(true, four_t2, t1, t0); });
    const long test_quadric = ops([&] { g_sink = needs_solve(gq_t2(gq, mc), gq_t1(gq, mc), gq_t0(gq, mc)); });
    const long test_linear = ops([&] { Counted t1 = lin_t1(lin, m), t0 = lin_t0(lin, m); g_sink = t1 > t0; });
    Counted t3, t2, t1, t0;
    // degree-3 surfaces: t3 .. t0 from the surface's Taylor data at the ray origin and the direction (rt_math.hpp, cubic_coefs), and
    // forming that data at a point (cubic_at) -- which happens once per (hit, wave that takes lights of its chunk), not per test
    CubicAt cat;
    // (with the error bounds cubic_guarded works with: cubic_mag_origin per point, cubic_mag_dir per test)
    const CubicAbs cab = cubic_abs(cub);
    CubicMag cmo, cmg;
    const long cubic_point = ops([&] { cat = cubic_at(cub, o); cmo = cubic_mag_origin(cab, o); });
    const long test_cubic_expand = ops([&] { cubic_coefs(cub, cat, d, t3, t2, t1, t0); cmg = cubic_mag_dir(cmo, fmax(fmax(fabs(d.x), fabs(d.y)), fabs(d.z))); });
    // the guarded solver on those coefficients, by the branch it takes (rt_math.hpp, cubic_guarded; a reciprocal or reciprocal square root
    // estimate counts as one operation, cbrt as a special function); "decide": the shadow-ray form that stops at the largest root
    const CubicMag tiny{1e-15, 1e-15, 1e-15, 1e-15};
    Counted tg;
    const long spg0 = g_special;
    const long guarded_cardano = ops([&] { g_sink = cubic_guarded(1.0, 0.0, 1.0, 1.0, tiny, 1e6, false, tg); });
    const long special_guarded_cardano = g_special - spg0;
    const long guarded_trig = ops([&] { g_sink = cubic_guarded(1.0, 0.0, -3.0, 1.0, tiny, 1e6, false, tg); });
    const long guarded_trig_decide = ops([&] { g_sink = cubic_guarded(1.0, 0.0, -3.0, 1.0, tiny, 1e6, true, tg); });
    const long guarded_quad = ops([&] { g_sink = cubic_guarded(0.0, 1.0, -10.0, 9.0, tiny, 1e6, false, tg); });
    const long guarded_linear = ops([&] { g_sink = cubic_guarded(0.0, 0.0, 2.0, -1.0, tiny, 1e6, false, tg); });
    const long test_cubic_dense = ops([&] { Mono mm; make_mono(mm, o, d); cubic_poly(cub, mm, t3, t2, t1, t0); }); // (the reference's expansion, term by term: the strict build)
    // --- root solves.  The deferred solve re-forms the coefficients from the table entry (second pass over the few candidates).
    auto quad_first = [&] { return ops([&] { g_sink = solve_quadlin(1.0, -10.0, 9.0) > 0.0; }); };   // first root accepted
    auto quad_second = [&] { return ops([&] { g_sink = solve_quadlin(1.0, 10.0, -9.0) > 0.0; }); };  // first root < EPS: second division
    const long qf = quad_first(); const DS qf_ds = last(); const long qs = quad_second(); const DS qs_ds = last();
    const double solve_quadlin_mean = 0.5 * (qf + qs);
    const DS solve_ds{0.5 * (qf_ds.div + qs_ds.div), 0.5 * (qf_ds.sqrt + qs_ds.sqrt)};
    const long recompute_us = ops([&] { Counted a = us_t1(us, m), b = us_t0(us, m); g_sink = a > b; });
    const long recompute_gq = ops([&] { Counted a = gq_t2(gq, mc), b = gq_t1(gq, mc), c = gq_t0(gq, mc); g_sink = a > b && b > c; });
    const long solve_linear = ops([&] { Counted t = -t0 / t1; g_sink = t > 0.0; });
    const DS lin_ds = last();
    const long sp0 = g_special;
    const long solve_cardano = ops([&] { g_sink = solve_cubic(1.0, 0.0, 1.0, 1.0) > 0.0; });          // delta > 0
    const DS cardano_ds = last();
    const long special_cardano = g_special - sp0;
    const long sp1 = g_special;
    const long solve_trig = ops([&] { g_sink = solve_cubic(1.0, 0.0, -3.0, 1.0) > 0.0; });            // three real roots
    const DS trig_ds = last();
    const long special_trig = g_special - sp1;
    // --- culling decisions (one lane each)
    const D3 org{0.1, 0.2, 0.3}, axis{0.0, 0.0, 1.0};
    const long cull_primary = ops([&] { g_sink = sphere_in_cone(us.kx, us.ky, us.kz, us.r, us.inv_r, org, axis, 0.999); });
    const DS cone_ds = last();
    TilePlanes P;
    const long tile_planes_ops = ops([&] { P = tile_planes(fa, -0.1, 0.1, -0.1, 0.1); });   // once per classifying lane (a tile and n_us / 4 spheres)
    const long cull_tile = ops([&] { g_sink = sphere_in_pyramid(us.kx, us.ky, us.kz, us.r, us.inv_r, org, P); });
    const Ball ball{1.0, 0.4, 8.0, 0.2};
    CullRec rec;
    const long cull_record_ops = ops([&] { rec = cull_record(us, ball); });
    const long cull_shadow_directional = ops([&] { g_sink = crec_relevant(rec, D3{dir_light.sdir[0], dir_light.sdir[1], dir_light.sdir[2]}, dir_light.inv_uu, dir_light.len_u); });
    const long cull_shadow_point = ops([&] { g_sink = sphere_relevant<true>(us, ball, pt_light); });
    const BoxH boxh{0.3, 0.2, 0.9, 0.0};
    const long cull_shadow_box = ops([&] { g_sink = crec_in_box_shadow(rec, boxh, D3{dir_light.sdir[0], dir_light.sdir[1], dir_light.sdir[2]}); });
    // --- rays
    const long primary_dir_ops = ops([&] { D3 dd = primary_dir_tab(fa, 0.05, -0.02); g_sink = dd.x > 0.0; });
    const DS pdir_ds = last();
    const long cone_axis_dot = ops([&] { g_sink = dot3(axis, d) > 0.0; });
    const long backface_dot = ops([&] { g_sink = dot3(d, D3{dir_light.p[0], dir_light.p[1], dir_light.p[2]}) > 0.0; });
    const long point_backface = ops([&] {   // rt_wavefront.hip phase B, point light: l - p, q = n.(l - p), |.| sum, FP32 round trip of the direction
        const Counted ex = pt_light.p[0] - o.x, ey = pt_light.p[1] - o.y, ez = pt_light.p[2] - o.z;
        const Counted q = dot3(d, D3{ex, ey, ez});
        const Counted mag = fabs(d.x * ex) + fabs(d.y * ey) + fabs(d.z * ez);
        g_sink = q < -1e-9 * mag;
    });
    const long hit_point = ops([&] { D3 sp{o.x + t1 * d.x, o.y + t1 * d.y, o.z + t1 * d.z}; g_sink = sp.x > 0.0; });
    const long normal_ops = ops([&] { D3 n = normal_vector(cub, o); g_sink = n.x > 0.0; });
    const DS normal_ds = last();
    const long sphere_normal_ops = ops([&] { D3 n = sphere_normal(us, o); g_sink = n.x > 0.0; }); // scenes of unit spheres only
    const long bias_ops = ops([&] { D3 b{o.x + SHADOW_BIAS * d.x, o.y + SHADOW_BIAS * d.y, o.z + SHADOW_BIAS * d.z}; g_sink = b.x > 0.0; });
    const long reflect_ops = ops([&] { D3 r = reflect_ray(d, D3{0.0, 1.0, 0.0}); g_sink = r.x > 0.0; });
    const F3 aop{0.2f, 0.2f, 0.2f};
    const float lc[3] = {1.0f, 1.0f, 1.0f};
    const long shade_directional = ops([&] { F3 c = surface_color_pre(dir_light.p, lc, false, o, d, aop); g_sink = c.x > 0.0f; });
    const long shade_point = ops([&] { F3 c = surface_color_pre(pt_light.p, lc, true, o, d, aop); g_sink = c.x > 0.0f; });
    const DS shade_pt_ds = last();
    const long ball_ops = 3 + 3 + 5 + 1 + 2 + 1; // rt_wavefront.hip phase A': extents (3 sub), centre (3 add 3 mul -> 6; counted 3 + 3), diagonal (3 mul 2 add), sqrt, scale + bias

    std::printf("{\n \"source\": \"tools/count_flops.cpp: the kernel's math headers (rt_math.hpp, rt_wavefront_math.hpp) executed over an operation-counting scalar; "
                "add / sub / mul / div / sqrt = 1 each, no FMA (strict build), comparisons / negation / fabs / conversions free\",\n \"units\": {\n");
    std::printf("  \"test_unitsq\": %ld,\n  \"test_quadric\": %ld,\n  \"test_linear\": %ld,\n  \"test_cubic_expand\": %ld,\n  \"cubic_point\": %ld,\n  \"test_cubic_dense\": %ld,\n", test_unitsq, test_quadric, test_linear,
                test_cubic_expand, cubic_point, test_cubic_dense);
    std::printf("  \"solve_unitsq\": %.1f,\n  \"solve_quadric\": %.1f,\n  \"solve_linear\": %ld,\n", recompute_us + solve_quadlin_mean, recompute_gq + solve_quadlin_mean, solve_linear);
    std::printf("  \"cubic_cardano\": %ld,\n  \"cubic_trig\": %ld,\n  \"cubic_quadratic\": %.1f,\n  \"cubic_linear\": %ld,\n", solve_cardano, solve_trig, solve_quadlin_mean, solve_linear);
    std::printf("  \"cubic_guarded_cardano\": %ld,\n  \"cubic_guarded_trig\": %ld,\n  \"cubic_guarded_trig_decide\": %ld,\n  \"cubic_guarded_quadratic\": %ld,\n  \"cubic_guarded_linear\": %ld,\n", guarded_cardano,
                guarded_trig, guarded_trig_decide, guarded_quad, guarded_linear);
    (void) special_guarded_cardano;
    std::printf("  \"tile_planes\": %ld,\n", tile_planes_ops);
    std::printf("  \"cull_tile\": %ld,\n  \"cull_primary\": %ld,\n  \"cull_shadow_directional\": %ld,\n  \"cull_shadow_point\": %ld,\n  \"cull_record\": %ld,\n  \"cull_shadow_box\": %ld,\n", cull_tile, cull_primary,
                cull_shadow_directional, cull_shadow_point, cull_record_ops, cull_shadow_box);
    // the kernel tallies a box-stage evaluation as RT_BOX_EVAL_UNITS directional decisions (rt_wavefront.hip): the two must agree
    if (cull_shadow_box > 3 * cull_shadow_directional + 2 || cull_shadow_box < 3 * cull_shadow_directional - 2) {
        std::fprintf(stderr, "count_flops: box-stage evaluation is %ld operations, the kernel books it as 3 x %ld\n", cull_shadow_box, cull_shadow_directional);
        return 1;
    }
    std::printf("  \"primary_ray\": %ld,\n  \"primary_ray_cross\": %ld,\n", primary_dir_ops + mono_o + mono_d + mono_od + cone_axis_dot, primary_dir_ops + mono_o_x + mono_d_x + mono_od_x + cone_axis_dot);
    std::printf("  \"shadow_ray_considered_directional\": %ld,\n  \"shadow_ray_considered_point\": %ld,\n", backface_dot, point_backface);
    std::printf("  \"shadow_ray_traced_directional\": %ld,\n  \"shadow_ray_traced_directional_cross\": %ld,\n", mono_od + 1, mono_od_x + 1);
    std::printf("  \"shadow_ray_traced_point\": %ld,\n  \"shadow_ray_traced_point_cross\": %ld,\n", 3 + mono_d + mono_od + 1, 3 + mono_d_x + mono_od_x + 1);
    std::printf("  \"hit\": %ld,\n  \"hit_cross\": %ld,\n  \"hit_spheres_only\": %ld,\n", hit_point + normal_ops + 4 * (bias_ops + mono_o), hit_point + normal_ops + 4 * (bias_ops + mono_o_x),
                hit_point + sphere_normal_ops + 4 * (bias_ops + mono_o));
    std::printf("  \"shade_directional\": %ld,\n  \"shade_point\": %ld,\n", shade_directional, shade_point);
    std::printf("  \"reflect_ray\": %ld,\n  \"reflect_ray_cross\": %ld,\n", reflect_ops + bias_ops + mono_o + mono_d + mono_od, reflect_ops + bias_ops + mono_o_x + mono_d_x + mono_od_x);
    std::printf("  \"chunk_ball\": %ld\n },\n", ball_ops);
    // divisions and square roots inside the units above (each counted as ONE operation there).  The compiler expands an FP64 division
    // into ~12 and a square root into ~14 FP64 instruction-flops (v_rcp / v_rsq seed + Newton steps in FMAs); bench.py uses these
    // counts to put the table on the footing of the hardware's instruction counters when it cross-checks the two.
    std::printf(" \"div_sqrt\": {\n");
    std::printf("  \"solve_unitsq\": [%.1f, %.1f], \"solve_quadric\": [%.1f, %.1f], \"cubic_quadratic\": [%.1f, %.1f],\n", solve_ds.div, solve_ds.sqrt, solve_ds.div, solve_ds.sqrt, solve_ds.div, solve_ds.sqrt);
    std::printf("  \"solve_linear\": [%.1f, %.1f], \"cubic_linear\": [%.1f, %.1f], \"cubic_cardano\": [%.1f, %.1f], \"cubic_trig\": [%.1f, %.1f],\n", lin_ds.div, lin_ds.sqrt, lin_ds.div, lin_ds.sqrt,
                cardano_ds.div, cardano_ds.sqrt, trig_ds.div, trig_ds.sqrt);
    std::printf("  \"cull_primary\": [%.1f, %.1f], \"primary_ray\": [%.1f, %.1f], \"primary_ray_cross\": [%.1f, %.1f],\n", cone_ds.div, cone_ds.sqrt, pdir_ds.div, pdir_ds.sqrt, pdir_ds.div, pdir_ds.sqrt);
    std::printf("  \"hit\": [%.1f, %.1f], \"hit_cross\": [%.1f, %.1f], \"hit_spheres_only\": [%.1f, %.1f], \"shade_point\": [%.1f, %.1f], \"chunk_ball\": [0.0, 1.0]\n },\n", normal_ds.div, normal_ds.sqrt,
                normal_ds.div, normal_ds.sqrt, normal_ds.div, normal_ds.sqrt, shade_pt_ds.div, shade_pt_ds.sqrt);
    std::printf(" \"special_functions\": {\"cubic_cardano\": %ld, \"cubic_trig\": %ld},\n", special_cardano, special_trig);
    std::printf(" \"reference_dense\": {\"expansion\": 286, \"linear\": 1, \"quadratic_miss\": 4, \"quadratic_hit\": 8, \"cardano\": 26, \"trig\": 39, \"normal_vector\": 79,\n"
                "  \"source\": \"SURVEY.md 8(d): the reference's as-written count per intersect_ray call (include/surface_impl.h:21-155)\"}\n}\n");
    return 0;
}
