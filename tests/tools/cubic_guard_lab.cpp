// cubic_guard_lab.cpp -- CPU experiment behind rt_math.hpp's cubic_guarded (test infrastructure: it links the oracle).
// For every primary ray of a scene and every shadow ray of its hits it compares, per degree-3 object,
//   the oracle's intersect_ray (dense expansion + the reference's solver, oracle/rt_oracle.c)      -- the truth
//   with cubic_guarded on the Taylor coefficients (the kernel's own header, compiled for the host) -- the candidate,
// and counts: tests, tests the guard passes on to the dense path, and among the ones it answers itself the decisions that differ
// (t >= EPS for primary rays, EPS < t < max_t for shadow rays) and the accepted roots that differ by more than 1e-7 relative.
// Built and driven by tests/tools/cubic_guard_lab.py.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>

#define RT_CUB_LAB 1
#include "rt_math.hpp"
#include "../../oracle/rt_oracle.h"

using namespace rtm;

struct LabStats {
    uint64_t tests[2], fallback[2], decision_diff[2], value_diff[2]; // [0] primary, [1] shadow
    double worst_rel[2];
    uint64_t fb_reason[8];
    uint64_t blocks[2], blocks_refusing[2]; // 8 x 8 pixel blocks (x light, for shadow rays) with a test / with a refused test
};

static bool is_cubic(const double *c)
{
    for (int k = 0; k < 10; k++)
        if (c[k] != 0.0) return true;
    return false;
}

static bool one(const double *c, const CubicAbs &ab, const CubicAt &ca, const CubicMag &mo, const double o[3], const double d[3], double max_t, int kind, bool strict_gt,
                LabStats *st, int verbose)
{
    const double t_ref = orc_intersect_ray(c, o, d);
    double t3, t2, t1, t0;
    cubic_coefs(c, ca, D3{d[0], d[1], d[2]}, t3, t2, t1, t0);
    const double dmax = fmax(fmax(fabs(d[0]), fabs(d[1])), fabs(d[2]));
    const CubicMag mg = cubic_mag_dir(mo, dmax);
    double t = 0.0;
    st->tests[kind]++;
    if (!cubic_guarded(t3, t2, t1, t0, mg, max_t, strict_gt, t)) {
        st->fallback[kind]++;
        st->fb_reason[g_cub_why & 7]++;
        return true;
    }
    const bool acc_ref = strict_gt ? (t_ref > EPS && t_ref < max_t) : (t_ref >= EPS && t_ref < max_t);
    const bool acc = strict_gt ? (t > EPS && t < max_t) : (t >= EPS && t < max_t);
    if (acc != acc_ref) {
        st->decision_diff[kind]++;
        if (verbose) {
            double tc[4];
            int br;
            orc_intersect_ray_ex(c, o, d, tc, &br);
            fprintf(stderr, "decision differs (%s): ref %.17g (branch %d) ours %.17g  dense t3..t0 %.17g %.17g %.17g %.17g  taylor %.17g %.17g %.17g %.17g\n", kind ? "shadow" : "primary", t_ref,
                    br, t, tc[3], tc[2], tc[1], tc[0], t3, t2, t1, t0);
        }
        return false;
    }
    if (acc && !strict_gt) { // (shadow rays are asked for the decision only)
        const double rel = fabs(t - t_ref) / fabs(t_ref);
        if (rel > st->worst_rel[kind]) st->worst_rel[kind] = rel;
        if (rel > 1e-7) {
            st->value_diff[kind]++;
            if (verbose) fprintf(stderr, "value differs (%s): ref %.17g ours %.17g rel %.3g\n", kind ? "shadow" : "primary", t_ref, t, rel);
        }
    }
    (void) ab;
    return false;
}

extern "C" void lab_run(const orc_scene *scene, const double cam[16], LabStats *st, int verbose)
{
    const double org[3] = {cam[12], cam[13], cam[14]};
    const uint32_t bw = (scene->px_width + 7) / 8, bh = (scene->px_height + 7) / 8, nl = scene->n_lights + 1;
    unsigned char *blk = (unsigned char *) calloc((size_t) bw * bh * nl, 1); // bit 0: tested, bit 1: refused; slot 0 primary, 1 + l shadow of light l
    for (uint32_t y = 0; y < scene->px_height; y++) {
        for (uint32_t x = 0; x < scene->px_width; x++) {
            unsigned char *b = blk + ((size_t) (y / 8) * bw + x / 8) * nl;
            double d[3];
            orc_primary_dir(scene, cam, (int) x, (int) y, d);
            // nearest hit by the oracle's rule (src/update-cpu.cpp:50-56)
            int best = -1;
            double best_t = INFINITY;
            for (uint32_t k = 0; k < scene->n_objects; k++) {
                const double *c = scene->objects[k].c;
                const double t = orc_intersect_ray(c, org, d);
                if (t >= EPS && t < 1e6 && t < best_t) {
                    best_t = t;
                    best = (int) k;
                }
                if (is_cubic(c)) {
                    const CubicAbs ab = cubic_abs(c);
                    const CubicAt ca = cubic_at(c, D3{org[0], org[1], org[2]});
                    const CubicMag mo = cubic_mag_origin(ab, D3{org[0], org[1], org[2]});
                    b[0] |= 1 | (one(c, ab, ca, mo, org, d, 1e6, 0, false, st, verbose) ? 2 : 0);
                }
            }
            if (best < 0) continue;
            double p[3], n[3], so[3];
            for (int i = 0; i < 3; i++) p[i] = org[i] + best_t * d[i];
            orc_normal_vector(scene->objects[best].c, p, n);
            for (int i = 0; i < 3; i++) so[i] = p[i] + SHADOW_BIAS * n[i];
            for (uint32_t l = 0; l < scene->n_lights; l++) {
                float fd[3];
                double max_t;
                orc_shadow_ray(&scene->lights[l], p, fd, &max_t); // (direction and max_t from the surface point, the ray starts at so: src/update-cpu.cpp:62-66)
                const double sd[3] = {(double) fd[0], (double) fd[1], (double) fd[2]};
                for (uint32_t k = 0; k < scene->n_objects; k++) {
                    const double *c = scene->objects[k].c;
                    if (!is_cubic(c)) continue;
                    const CubicAbs ab = cubic_abs(c);
                    const CubicAt ca = cubic_at(c, D3{so[0], so[1], so[2]});
                    const CubicMag mo = cubic_mag_origin(ab, D3{so[0], so[1], so[2]});
                    b[1 + l] |= 1 | (one(c, ab, ca, mo, so, sd, max_t, 1, true, st, verbose) ? 2 : 0);
                }
            }
        }
    }
    for (size_t i = 0; i < (size_t) bw * bh; i++)
        for (uint32_t l = 0; l < nl; l++) {
            const unsigned char v = blk[i * nl + l];
            if (v & 1) st->blocks[l ? 1 : 0]++;
            if (v & 2) st->blocks_refusing[l ? 1 : 0]++;
        }
    free(blk);
}

// cubic_guarded itself, for unit tests: returns 1 (answered; *t = the root the callers compare) or 0 (refused).  m[4] = the uncertainties m3 .. m0.
extern "C" int lab_guard(double t3, double t2, double t1, double t0, const double m[4], double max_t, int decide, double *t)
{
    double r = 0.0;
    const bool ok = cubic_guarded(t3, t2, t1, t0, CubicMag{m[0], m[1], m[2], m[3]}, max_t, decide != 0, r);
    *t = r;
    return ok ? 1 : 0;
}
// the reference's solver on the same coefficients (rt_math.hpp: solve_cubic / solve_quadlin, as intersect_cubic chains them)
extern "C" double lab_reference(double t3, double t2, double t1, double t0)
{
    return fabs(t3) > EPS ? solve_cubic(t3, t2, t1, t0) : solve_quadlin(t2, t1, t0);
}
