/*
 * okenv_math.h -- scalar math shared, bit for bit, by the HIP kernels and the CPU oracle.
 *
 * Why this exists (SURVEY.md section 0 item 5, section 7 "Transcendentals"): the reference's host
 * code calls glibc cosf/sinf (Environment/Agent.cpp:94-97,115-118, Environment/CollisionChecker.cu:
 * 121-124,157-158) and its kernel calls CUDA cosf/sinf (Environment/CollisionChecker.cu:47-48).
 * Those differ from each other and from ROCm's OCML in the last ulp, which is enough to flip the
 * `min_dist2 < 2.0f` crash test.  Crash/done flags must be bit-exact between the GPU path and the
 * CPU oracle, so both sides evaluate sine/cosine through THIS header: an fp64 reduction and fp64
 * polynomials, rounded once to fp32.  Only +, -, *, explicitly requested fused multiply-adds (OK_FMA) and
 * round-to-nearest-even conversions are used, all IEEE-exact on x86-64 and on gfx950, so the two sides agree bit
 * for bit provided the translation unit is compiled with -ffp-contract=off (nothing fused that does not ask for it).
 * tests/test_math.py bounds the distance to glibc sincosf (<= 1 ulp) and to an fp64 reference.
 *
 * Also here: Philox4x32-10 (Salmon et al., SC'11), the counter-based generator the C2 bench recipe
 * draws actions and reset positions from, so the device rollout and the oracle
 * rollout see identical streams.
 *
 * Plain C99 / C++ / HIP.  No dependency on anything under oracle/.
 */
#ifndef OKENV_MATH_H
#define OKENV_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define OK_HD __host__ __device__ static inline
#else
#define OK_HD static inline
#endif

/* Environment/Typedefs.h:10  kDeg2Rad = float(M_PI / 180.0F)  (bit pattern 0x3C8EFA35) */
#define OK_DEG2RAD 0.01745329238474369049072265625f
/* Environment/Agent.cpp:84,110  kDt{0.016} as float (bit pattern 0x3C83126F) */
#define OK_DT 0.016000000759959220886230468750f
/* Environment/Agent.h:10-11 */
#define OK_SENSOR_RANGE 200.0f
#define OK_SPEED_LIMIT 100.0f
/* Environment/CollisionChecker.cu:167 */
#define OK_CRASH_DIST2 2.0f
/* Environment/Environment.h:19-20 */
#define OK_DISP_PERIOD 200u
#define OK_DISP_THRESH2 400.0f /* kDisplamentThreshold^2 = 20*20, exact in fp32 */
/* Environment/CollisionChecker.cu:23 */
#define OK_PARALLEL_EPS 1e-8f

#define OK_RINT(x) __builtin_rint(x) /* round to nearest even: rint() on the host, v_rndne_f64 on gfx950 */
/* a * b + c with ONE rounding, asked for explicitly: v_fma_f64 on gfx950, libm's correctly rounded fma() (or the host's FMA
 * instruction) on x86-64 -- the same bits either way.  The translation units are still compiled -ffp-contract=off: nothing is
 * fused that does not say so here. */
#define OK_FMA(a, b, c) __builtin_fma((a), (b), (c))
/* A double constant of the polynomials below.  On the device it is made opaque and pinned to a scalar register pair: hipcc would
 * otherwise materialise each one in a VECTOR register pair (v_fmac_f64 wants its addend there), hoist all of them out of the
 * kernels' step loops and keep them alive across the raycast -- 22 more VGPRs in every step kernel, scratch spills in the policy
 * ones (profiles/r4/kernel_resource_usage.txt).  As scalar operands of v_fma_f64 they cost s_mov's and no vector registers.  The
 * value is untouched: same bits on both sides. */
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ inline __attribute__((always_inline)) double ok_konst(double c)
{
    asm("" : "+s"(c));
    return c;
}
#else
#define ok_konst(c) (c)
#endif

/*
 * sin and cos of an fp32 angle [rad], from an fp64 evaluation rounded once to fp32 (round 4's form; rounds 1-3 used a
 * three-part Cody-Waite reduction and Taylor series to 1/15! with separate multiplications and additions: ~42 fp64 operations
 * in a dependent chain of ~26; this is 21 operations in a chain of 12).
 *   q = rint(x * 2/pi);  r = x - q * pi/2 by two FMAs against pi/2 = P1 + P2 (P1 the nearest double, P2 the next 53 bits): each FMA
 *   forms its product exactly and rounds once, so r carries a RELATIVE error of ~2^-52 however close x lies to a multiple of
 *   pi/2, for every |q| < 2^31;
 *   sin r = r + (r z) S(z),  cos r = 1 + z C(z),  z = r^2, S and C of degree 5 in z: interpolants at the Chebyshev nodes of
 *   [0, 0.79^2] (tools/fit_math.py prints them; pi/4 = 0.7854, the slack covers q's own rounding), relative errors 2^-55 and
 *   2^-51, evaluated by Horner's rule in FMAs.
 * The result is the correctly rounded fp32 sine / cosine but for about one argument in 10^7 (tests/test_math.py: equal to the
 * rounded fp64 value on 700 000 samples and on the floats nearest to multiples of pi/2 up to 2^31; <= 1 ulp from glibc's
 * sinf / cosf, which the reference's host code calls).  rot_ is never wrapped (Environment/Agent.cpp:86,112), so large arguments
 * do occur: beyond 2^31 rad (fp32 spacing there is 256 rad: the angle carries no information) the argument is folded by an
 * exact fmod first; NaN / Inf give NaN.
 */
OK_HD void ok_sincosf(float x, float *s_out, float *c_out)
{
    double xd = (double)x;
    if (!(__builtin_fabs(xd) < 2147483648.0)) {
        xd = __builtin_fmod(xd, 6.283185307179586476925286766559); /* exact by definition: identical on CPU and GPU */
        if (!(xd == xd)) { /* NaN or Inf in */
            *s_out = (float)xd;
            *c_out = (float)xd;
            return;
        }
    }
    const double q = OK_RINT(xd * ok_konst(0x1.45f306dc9c883p-1)); /* 2/pi */
    double r = OK_FMA(-q, ok_konst(0x1.921fb54442d18p+0), xd);     /* P1 = 1.5707963267948966    */
    r = OK_FMA(-q, ok_konst(0x1.1a62633145c07p-54), r);            /* P2 = 6.123233995736766e-17 */
    const int n = ((int)q) & 3;                          /* quadrant; |q| < 1.4e9 fits an int */
    const double z = r * r;
    double ps = ok_konst(0x1.5e01d1798c2b3p-33);              /*  1.5916480269048027e-10 */
    ps = OK_FMA(ps, z, ok_konst(-0x1.ae5ff116c8d06p-26));     /* -2.505110882200661e-08  */
    ps = OK_FMA(ps, z, ok_konst(0x1.71de377d84985p-19));      /*  2.755731599143529e-06  */
    ps = OK_FMA(ps, z, ok_konst(-0x1.a01a019e70424p-13));     /* -1.9841269836543094e-04 */
    ps = OK_FMA(ps, z, ok_konst(0x1.1111111110b60p-7));       /*  8.333333333330806e-03  */
    ps = OK_FMA(ps, z, ok_konst(-0x1.5555555555555p-3));      /* -1.6666666666666666e-01 */
    const double sr = OK_FMA(r * z, ps, r);
    double pc = ok_konst(0x1.1bfd9695386eap-29);              /*  2.0663034153592203e-09 */
    pc = OK_FMA(pc, z, ok_konst(-0x1.27e0dd327adbcp-22));     /* -2.75558210165445e-07   */
    pc = OK_FMA(pc, z, ok_konst(0x1.a019fc4c6ed87p-16));      /*  2.4801582456863223e-05 */
    pc = OK_FMA(pc, z, ok_konst(-0x1.6c16c168f930cp-10));     /* -1.3888888881805088e-03 */
    pc = OK_FMA(pc, z, ok_konst(0x1.5555555554001p-5));       /*  4.166666666662878e-02  */
    pc = OK_FMA(pc, z, ok_konst(-0x1.ffffffffffffap-2));      /* -4.9999999999999967e-01 */
    const double cr = OK_FMA(z, pc, 1.0);
    double sv, cv;
    if (n == 0) { sv = sr; cv = cr; }
    else if (n == 1) { sv = cr; cv = -sr; }
    else if (n == 2) { sv = -sr; cv = -cr; }
    else { sv = -cr; cv = sr; }
    *s_out = (float)sv;
    *c_out = (float)cv;
}

/* ---- Philox4x32-10 -------------------------------------------------------------------------- */

typedef struct ok_u32x4 { uint32_t v[4]; } ok_u32x4;

OK_HD ok_u32x4 ok_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
    for (int round = 0; round < 10; ++round) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    ok_u32x4 r;
    r.v[0] = c0; r.v[1] = c1; r.v[2] = c2; r.v[3] = c3;
    return r;
}

/* u32 -> [0,1) with 24 random bits; every step is exact in fp32 */
OK_HD float ok_u01(uint32_t u)
{
    return (float)(u >> 8) * 5.9604644775390625e-08f; /* 2^-24 */
}

/*
 * The C2 synthetic-action recipe:
 *   counter = (agent, step, 0, 0), key = (seed, 0x6F6B656E /"oken"/)
 *   throttle = U[0,100), steer = U[-5,5), reset_idx = floor(U * P) drawn from the third word.
 * `agent` is the GLOBAL agent id, so a population sharded over ranks reproduces the unsharded streams.
 */
typedef struct ok_random_action { float throttle; float steer; uint32_t reset_word; } ok_random_action;

OK_HD ok_random_action ok_draw_random_action(uint32_t seed, uint32_t agent, uint32_t step)
{
    const ok_u32x4 r = ok_philox4x32(agent, step, 0u, 0u, seed, 0x6F6B656Eu);
    ok_random_action a;
    a.throttle = ok_u01(r.v[0]) * 100.0f;
    a.steer = ok_u01(r.v[1]) * 10.0f - 5.0f;
    a.reset_word = r.v[2];
    return a;
}

/* index in [0,P) from a 32-bit word (multiply-shift, no modulo bias worth speaking of) */
OK_HD uint32_t ok_index_from_word(uint32_t w, uint32_t P)
{
    return (uint32_t)(((uint64_t)w * (uint64_t)P) >> 32);
}

/* start index of agent j in the bench recipe: (j * 2654435761) mod 2^32 mod P */
OK_HD uint32_t ok_start_index(uint32_t agent, uint32_t P)
{
    return (uint32_t)(agent * 2654435761u) % P;
}

/* ---- Environment::resetAgent (Environment/Environment.cpp:79-122) -- */

/* the three booleans of resetAgent(agent, pick_random_point, randomize_lane, randomize_heading) */
#define OK_RESET_RANDOM_POINT 1u
#define OK_RESET_RANDOM_LANE 2u
#define OK_RESET_RANDOM_HEADING 4u
/* batch calls only: leave agents whose crashed_ flag is clear alone */
#define OK_RESET_ONLY_DONE 8u
/* RaceTrack::kStartingIdx (Environment/RaceTrack.h:18) */
#define OK_RESET_START_IDX 3u

/*
 * The random draws of one resetAgent call.  raylib's GetRandomValue(lo, hi) (inclusive integer range, un-vendored,
 * global state) is replaced by one Philox block per (agent, epoch):
 *   counter = (agent, epoch, 1, 0), key = (seed, 0x6F6B656E)
 *   word 0 -> reset_idx in [0, P-1]            (Environment.cpp:76)
 *   word 1 -> heading draw in [0, 45]          (:92)
 *   word 2 -> lane draw in [10, 90], / 100.F   (:111)
 * `ctr` stands for the reference's function-static call counter (:88): even calls turn the heading offset negative.
 * The draws the reference skips (flags off) are simply not used, so one flag does not shift another's stream.
 */
typedef struct ok_reset_draw { uint32_t idx; float heading_offset; float alpha; int use_lane; } ok_reset_draw;

OK_HD ok_reset_draw ok_draw_reset(uint32_t seed, uint32_t agent, uint32_t epoch, uint32_t ctr, uint32_t P, uint32_t flags)
{
    const ok_u32x4 r = ok_philox4x32(agent, epoch, 1u, 0u, seed, 0x6F6B656Eu);
    const int random_point = (flags & OK_RESET_RANDOM_POINT) != 0u;
    ok_reset_draw d;
    d.idx = random_point ? ok_index_from_word(r.v[0], P) : OK_RESET_START_IDX;
    d.heading_offset = 0.0f;
    if (random_point && (flags & OK_RESET_RANDOM_HEADING) != 0u) {
        const float kHeadingRangeDeg = 45.0f;
        const float draw = (float)ok_index_from_word(r.v[1], 46u);
        d.heading_offset = (ctr % 2u == 0u) ? (draw + kHeadingRangeDeg) * -1.0f : draw + kHeadingRangeDeg;
    }
    d.use_lane = random_point && (flags & OK_RESET_RANDOM_LANE) != 0u;
    d.alpha = d.use_lane ? (float)(10u + ok_index_from_word(r.v[2], 81u)) / 100.0f : 0.0f;
    return d;
}

/* Pose of the reset (Environment.cpp:104-121): a point between the inner lane boundaries or the centre-line point,
 * track heading plus the offset.  `li`/`ri` are left_bound_inner_/right_bound_inner_ as xy pairs. */
OK_HD void ok_reset_pose(const ok_reset_draw d, const float *cx, const float *cy, const float *chead, const float *li,
                         const float *ri, float *x, float *y, float *rot)
{
    if (d.use_lane) {
        const float lx = li[2u * d.idx], ly = li[2u * d.idx + 1u];
        const float rx = ri[2u * d.idx], ry = ri[2u * d.idx + 1u];
        *x = lx * d.alpha + rx * (1.0f - d.alpha);
        *y = ly * d.alpha + ry * (1.0f - d.alpha);
    } else {
        *x = cx[d.idx];
        *y = cy[d.idx];
    }
    *rot = chead[d.idx] + d.heading_offset;
}

/* ---- EvolutionaryRacer policy ------------------------------- */

/* Weight layout per agent (floats): w1[(R+2)][OK_MLP_HID_PAD] then w2[OK_MLP_HID_PAD][OK_MLP_OUT_PAD]; entries beyond
 * the hidden width H (30 in the reference, EvolutionaryRacer/Network.hpp:92-95) and beyond the 6 outputs are zero. */
#define OK_MLP_HID_PAD 32
#define OK_MLP_OUT 6
#define OK_MLP_OUT_PAD 8
#define OK_MLP_WEIGHTS(R) (((R) + 2) * OK_MLP_HID_PAD + OK_MLP_HID_PAD * OK_MLP_OUT_PAD)

/* genetic::normalizeAngleDeg (EvolutionaryRacer/Network.hpp:16-27): repeated +-360 in fp32, bit for bit; the loops
 * are capped at 65536 turns each so that an infinite rot_ cannot hang a kernel (the reference would spin forever). */
OK_HD float ok_normalize_angle_deg(float angle)
{
    for (int i = 0; i < 65536 && angle < 360.0f; ++i) angle += 360.0f;
    for (int i = 0; i < 65536 && angle >= 360.0f; ++i) angle -= 360.0f;
    return angle;
}

/* tanh for the CMA-ES controller (CovarianceMatrixAdaptationEvolution/Controller.cpp:16-23), evaluated in fp64 with IEEE operations
 * and explicit FMAs only (no library call), so that the device and the CPU oracle produce the same bits -- the same idea as
 * ok_sincosf.  tanh|x| = (1 - e) / (1 + e), e = exp(-2|x|) = 2^n (1 + m), n = rint(-2|x| / ln 2), m = expm1(r), |r| <= ln 2 / 2,
 * r by two FMAs against ln 2 = L1 + L2, m = r + r^2 E(r) with E of degree 8 (interpolant at the Chebyshev nodes of +-0.347,
 * tools/fit_math.py: relative error 2^-48).  Numerator and denominator are formed from m, not from e:
 * 1 - e = (1 - 2^n) - 2^n m and 1 + e = (1 + 2^n) + 2^n m, one FMA each with 1 -+ 2^n exact -- for small |x| (n = 0) the numerator
 * is -m itself, so nothing cancels and no separate small-argument branch is needed beyond the one below.  (Rounds 1-3: Taylor
 * series of exp to r^13 by separate multiplications and additions, 1 - e by subtraction.)  tests/test_math.py: equal to the
 * rounded fp64 tanh on all but a handful of a million samples; glibc's tanhf, not correctly rounded itself, is within 2 ulps. */
OK_HD float ok_tanhf(const float x)
{
    const double ax = __builtin_fabs((double)x);
    if (!(ax >= 0.000244140625)) /* |x| < 2^-12: x^3/3 is below half an ulp of x; NaN takes this exit too and stays NaN */
        return x;
    const double y = -2.0 * (ax < 20.0 ? ax : 20.0); /* tanh(20) is 1 to 17 digits: larger arguments change nothing */
    const double n = OK_RINT(y * ok_konst(0x1.71547652b82fep+0)); /* 1 / ln 2 */
    double r = OK_FMA(-n, ok_konst(0x1.62e42fefa39efp-1), y);     /* L1 = 0.6931471805599453     */
    r = OK_FMA(-n, ok_konst(0x1.abc9e3b39803fp-56), r);           /* L2 = 2.3190468138462996e-17 */
    double p = ok_konst(0x1.28809b1a1156ep-22);           /* 2.7613934750302004e-07 */
    p = OK_FMA(p, r, ok_konst(0x1.72c7b3ac3a215p-19));    /* 2.7625269095508424e-06 */
    p = OK_FMA(p, r, ok_konst(0x1.a019c964743c8p-16));    /* 2.4801536157973374e-05 */
    p = OK_FMA(p, r, ok_konst(0x1.a019ad41ef162p-13));    /* 1.9841208455585307e-04 */
    p = OK_FMA(p, r, ok_konst(0x1.6c16c1739cf9cp-10));    /* 1.388888890599716e-03  */
    p = OK_FMA(p, r, ok_konst(0x1.1111111c5b16fp-7));     /* 8.333333353868181e-03  */
    p = OK_FMA(p, r, ok_konst(0x1.5555555554ca3p-5));     /* 4.166666666665122e-02  */
    p = OK_FMA(p, r, ok_konst(0x1.5555555553b3cp-3));     /* 1.6666666666648122e-01 */
    p = OK_FMA(p, r, 0.5);
    const double m = OK_FMA(r * r, p, r); /* expm1(r) */
    union { uint64_t u; double d; } two_n; /* 2^n, n in [-58, 0] */
    two_n.u = (uint64_t)(1023 + (int)n) << 52;
    const double num = OK_FMA(-two_n.d, m, 1.0 - two_n.d);
    const double den = OK_FMA(two_n.d, m, 1.0 + two_n.d);
    const double t = num / den;
    return (float)(x < 0.0f ? -t : t);
}

/* One candidate's controller (Controller.cpp:3-23: fc1 in->h, fc2 h->h/2, fc3 h/2->out, tanh after each) on the input
 * CmaEsAgent::stateToTensor builds (main_eigen.cpp:45-56: ||sensor_hits_[i]|| / kSensorRange).  `params` in the order of
 * torch's parameters(): fc1.weight [h][in] row-major, fc1.bias [h], fc2.weight [h/2][h], fc2.bias, fc3.weight [out][h/2],
 * fc3.bias (Controller.cpp:36-53).  A unit's sum starts from its bias and adds w * x in ascending input order, fp32, no
 * FMA (libtorch's own order inside addmm is not specified): okControllerKernel on the device, ctrl_forward in the oracle.
 * h <= 64. */
#define OK_CTRL_MAX_HIDDEN 64
OK_HD int ok_controller_num_params(const int in, const int hidden, const int out)
{
    const int h2 = hidden / 2;
    return hidden * in + hidden + h2 * hidden + h2 + out * h2 + out;
}
/* nn_output > kOutputActivationLim (GeneticAgent.hpp:45-54) on a PRE-activation z, where nn_output = sigmoid(z) =
 * 1.F / (1.F + exp(-z)) in fp32 (Network.hpp:162-165).  The quotient exceeds 0.5 exactly when the rounded sum 1 + exp(-z)
 * is below 2, i.e. when exp(-z) rounds to 1 - 2^-23 or less, i.e. when exp(-z) <= 1 - 1.5 * 2^-24 (the tie goes to the even
 * neighbour, 1 - 2^-23), i.e. when z >= -ln(1 - 1.5 * 2^-24) = 1.5 * 2^-24 + 4.0e-15: the first float at or above that is
 * 0x33C00001.  So the test is one comparison, with the bits of an fp32 sigmoid whose exp is correctly rounded there (glibc's
 * expf is: tests/test_math.py compares the two over every float around the threshold).  For 0 < z < 8.94e-8 the sigmoid is
 * exactly 0.5 and the output is NOT active. */
OK_HD int ok_sigmoid_above_half(const float z)
{
    union { uint32_t u; float f; } t;
    t.u = 0x33C00001u;
    return z >= t.f; /* false for NaN, like the comparison with the sigmoid */
}

/* GeneticAgent::updateAction's decode (EvolutionaryRacer/GeneticAgent.hpp:45-54) from the six pre-activations z. */
OK_HD void ok_ga_decode_action(const float z[OK_MLP_OUT], float *throttle, float *steer)
{
    float t = 0.0f, s = 0.0f;
    t += ok_sigmoid_above_half(z[0]) ? 0.3f : 0.0f;
    t += ok_sigmoid_above_half(z[1]) ? -0.3f : 0.0f;
    s += ok_sigmoid_above_half(z[2]) ? 1.0f : 0.0f;
    s += ok_sigmoid_above_half(z[3]) ? 4.0f : 0.0f;
    s += ok_sigmoid_above_half(z[4]) ? -1.0f : 0.0f;
    s += ok_sigmoid_above_half(z[5]) ? -4.0f : 0.0f;
    *throttle = t;
    *steer = s;
}

/* Deterministic stand-ins for the reference's unseeded generators (Eigen Random(), std::random_device):
 * initial weight w of agent a:      U[-1,1) from Philox(counter = (a, w, 1, 0), key = (seed, "oken"))
 * mating draws of offspring o, weight w in generation g: Philox(counter = (o, w, 2, g)) -> u_mutate, u_value, u_parent
 * parent choice of offspring o in generation g:          Philox(counter = (o, try, 3, g)) */
OK_HD float ok_ga_initial_weight(uint32_t seed, uint32_t agent, uint32_t w)
{
    const ok_u32x4 r = ok_philox4x32(agent, w, 1u, 0u, seed, 0x6F6B656Eu);
    return ok_u01(r.v[0]) * 2.0f - 1.0f;
}

/* Parent choice (EvolutionaryRacer/Mating.hpp:128-152): offspring 0 clones the best, offspring 1 mates the best with
 * itself, every other offspring draws two DIFFERENT parents with probability proportional to the parents' scores
 * (std::discrete_distribution; uniform when every score is zero).  ps[] = scores of the K best agents, best first.
 * Returns first | second << 8, bit 16 = exact clone. */
OK_HD uint32_t ok_ga_pick_parent(const float *ps, int K, float u)
{
    float total = 0.0f;
    for (int k = 0; k < K; ++k) total += (ps[k] > 0.0f ? ps[k] : 0.0f);
    if (!(total > 0.0f)) {
        const int k = (int)(u * (float)K);
        return (uint32_t)(k < K ? k : K - 1);
    }
    const float x = u * total;
    float acc = 0.0f;
    for (int k = 0; k < K; ++k) {
        acc += (ps[k] > 0.0f ? ps[k] : 0.0f);
        if (x < acc) return (uint32_t)k;
    }
    return (uint32_t)(K - 1);
}

OK_HD uint32_t ok_ga_parent_pair(const float *ps, int K, uint32_t seed, uint32_t offspring, uint32_t generation)
{
    if (offspring == 0u) return 0u | (0u << 8) | (1u << 16); /* clone of the best */
    if (offspring == 1u || K < 2) return 0u | (0u << 8);     /* best mated with itself: mutations only */
    const ok_u32x4 r0 = ok_philox4x32(offspring, 0u, 3u, generation, seed, 0x6F6B656Eu);
    const uint32_t first = ok_ga_pick_parent(ps, K, ok_u01(r0.v[0]));
    uint32_t second = first;
    for (uint32_t attempt = 1u; attempt <= 16u && second == first; ++attempt) {
        const ok_u32x4 r = ok_philox4x32(offspring, attempt, 3u, generation, seed, 0x6F6B656Eu);
        second = ok_ga_pick_parent(ps, K, ok_u01(r.v[0]));
    }
    if (second == first) second = (first + 1u) % (uint32_t)K; /* all but impossible; keeps "two different parents" */
    return first | (second << 8);
}

/* true for the real entries of the padded per-agent weight block (index i in [0, OK_MLP_WEIGHTS(R))) */
OK_HD int ok_mlp_weight_is_real(uint32_t i, int R, int H)
{
    const uint32_t n1 = (uint32_t)((R + 2) * OK_MLP_HID_PAD);
    if (i < n1) return (int)((i % OK_MLP_HID_PAD) < (uint32_t)H);
    const uint32_t q = i - n1;
    return (int)((q / OK_MLP_OUT_PAD) < (uint32_t)H && (q % OK_MLP_OUT_PAD) < OK_MLP_OUT);
}

/* ---- RLRacers/Q_Learning ------------------------------------------------ */

#define OK_Q_STATES 243 /* 5 rays x 3 proximity bins, QAgent.hpp:31-34 */
#define OK_Q_ACTIONS 3
#define OK_Q_INVALID (-3.40282346638528859811704183484516925e+38f) /* numeric_limits<float>::lowest(), QAgent.hpp:36 */

/* QLearnAgent::discretizeState's bin of one ray (QAgent.hpp:72-94): < 5 -> 0, < 10 -> 1, else 2 */
OK_HD int ok_q_bin(float ray_dist)
{
    return (ray_dist < 5.0f) ? 0 : ((ray_dist < 10.0f) ? 1 : 2);
}

/* index of the largest of three values, the first on ties (std::max_element, QAgent.hpp:108-110) */
OK_HD int ok_q_argmax3(float q0, float q1, float q2)
{
    int idx = 0;
    float m = q0;
    if (q1 > m) { m = q1; idx = 1; }
    if (q2 > m) { idx = 2; }
    return idx;
}

/* Epsilon-greedy draw (QAgent.hpp:98-119) with Philox(counter = (agent, step, 4, 0)) standing in for raylib's
 * GetRandomValue: explore iff u0 < epsilon, random action = min(2, floor(3 u1)). */
OK_HD int ok_q_choose_action(uint32_t seed, uint32_t agent, uint32_t step, float epsilon, float q0, float q1, float q2)
{
    const ok_u32x4 r = ok_philox4x32(agent, step, 4u, 0u, seed, 0x6F6B656Eu);
    if (ok_u01(r.v[0]) < epsilon) {
        const int a = (int)(ok_u01(r.v[1]) * 3.0f);
        return a > 2 ? 2 : a;
    }
    return ok_q_argmax3(q0, q1, q2);
}

/* The part of that draw that does not depend on the Q values: -1 = exploit (take the argmax), else the random action.  For
 * callers that make many steps' draws at once (okQSettleKernel); ok_q_choose_action == (d < 0 ? argmax3 : d). */
OK_HD int ok_q_draw_action(uint32_t seed, uint32_t agent, uint32_t step, float epsilon)
{
    const ok_u32x4 r = ok_philox4x32(agent, step, 4u, 0u, seed, 0x6F6B656Eu);
    if (ok_u01(r.v[0]) < epsilon) {
        const int a = (int)(ok_u01(r.v[1]) * 3.0f);
        return a > 2 ? 2 : a;
    }
    return -1;
}

/* kActionMap (QAgent.hpp:40-42): 0 -> (60, 0), 1 -> (30, +5), 2 -> (30, -5) */
OK_HD void ok_q_action_values(int action, float *throttle, float *steer)
{
    *throttle = (action == 0) ? 60.0f : 30.0f;
    *steer = (action == 0) ? 0.0f : ((action == 1) ? 5.0f : -5.0f);
}

/* QLearnAgent::reward (QAgent.hpp:150-168); *prev_idx is updated only when the agent has not crashed */
OK_HD float ok_q_reward(int crashed, int nearest_idx, int *prev_idx, int track_len)
{
    if (crashed) return -200.0f;
    int progression = nearest_idx - *prev_idx;
    *prev_idx = nearest_idx;
    if (progression < 0) progression = -progression;
    return (float)((progression > track_len / 2) ? track_len - progression : progression);
}

/* QLearnAgent::learn (QAgent.hpp:121-138): returns the new Q(s, a) */
OK_HD float ok_q_learn(float old_q, float max_q_next, float reward)
{
    const float target = reward + 0.8f * max_q_next;
    if (old_q == OK_Q_INVALID || max_q_next == OK_Q_INVALID) return reward;
    return old_q + 0.2f * (target - old_q);
}

#endif /* OKENV_MATH_H */
