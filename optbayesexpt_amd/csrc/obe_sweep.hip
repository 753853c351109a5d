// K1 + K5 — the utility sweep: model evaluated over the settings x draws grid, reduced to
// the per-setting variance of the predicted output, then utility and argmax
// (obe_base.py:463-489 yvar_from_parameter_draws, :628-655 utility_variance,
// :733-756 opt_setting).
//
// Mapping (gfx950):
//   * lane <-> setting.  Each thread owns SPT settings in registers (prepared setting,
//     shift, and the two running moments per channel); nothing is reduced across lanes
//     inside the loop.
//   * sweep_pack_kernel packs every particle ONCE per sweep (per-particle divisions and
//     sqrt(w) hoisted out of the grid) into an AoS array in the workspace.
//   * the particle axis is streamed through the SCALAR path: a packed particle is the same
//     for all 64 lanes, so it is fetched with s_load_dwordx8 into SGPRs (one group ahead of
//     its use) and enters the FP64 instructions as their scalar operand — no LDS tile, no
//     barrier, no VGPRs for broadcast values, and each wave runs on its own.
//   * a workgroup is 4 waves that own the SAME 64*SPT settings and a quarter each of one
//     particle chunk; their moments are added in a fixed order through LDS at the end, so a
//     work item is (64*SPT settings) x (one chunk): ~5000 of them at 65 536 x 1 048 576, six
//     per resident workgroup slot (the launch is one wave of equal items, so its duration is
//     set by how evenly they finish), with partials of only n_chunks x N_s.
//   * grid = setting tiles x particle chunks, XCD-aware; chunk partial moments share one
//     shift per setting, so they simply add (finalize pass).
//   * arithmetic: FP64 VALU only (no contraction over a shared operand => no MFMA).
//     The roof is the FP64 vector rate, not HBM: compulsory traffic is
//     8(D+1)N_p + 8 S N_s bytes for N_s*N_p evaluations.
//
// Variance numerics: moments are accumulated about a per-setting shift
// c_s = model(x_s; mean parameters), so  var = (S2 - S1^2/W)/W  does not cancel
// catastrophically (np.var is two-pass; the shift plays the role of its first pass).
#include <cstdlib>
#include <cstring>
#include <ctime>

#include "obe_common.h"
#include "obe_models.h"

namespace obe {

constexpr int kSweepWaves = kBlock / kWave;   // waves of a workgroup: same settings, a quarter of the chunk each
constexpr int kMaxChunks = 1024;
constexpr int kFinMaxBlocks = 65536;     // finalize workgroups (64 settings each) the argmax partials hold
constexpr int kFinSettings = kWave;      // settings per finalize workgroup

static int64_t argmax_slots(int64_t n_settings) {
    return std::max<int64_t>(kMaxBlocks, (n_settings + kFinSettings - 1) / kFinSettings);
}

// one-workgroup path for reference-semantics sweeps (sweep_small_kernel)
constexpr int64_t kSmallSweepDraws = 256, kSmallSweepEvals = 131072;

// packed particle: M::NPK doubles + sqrt(weight), padded to 16 bytes
template <class M>
constexpr int packed_width() { return (M::NPK + 2) & ~1; }
#ifdef OBE_PLUGIN_MODEL_HEADER
constexpr int kMaxPackedWidth = packed_width<PluginModel>();
#else
// every registry model: Lorentz<K> K + 2 <= n_dims, lines <= 2, Rabi 3, Coil 7
static int max_packed_width(int n_dims) { return std::max(8, (n_dims + 2) & ~1); }
#endif

struct SweepPlan {
    int spt;           // settings per thread
    int tiles_x;       // setting tiles of 64 * spt
    int nchunks;       // particle chunks
    int64_t chunk;     // draws per chunk
    int nchunks_bound; // >= nchunks of this plan and of every plan with fewer draws (workspace sizing)
};

// Tuned on MI355X at 65 536 x 1 048 576 (tools/exp_sweep_launch.sh): 8 settings per lane (one
// batched reciprocal per 8 evaluations); 768 workgroups are resident (3 per CU at <= 168 VGPRs),
// and the launch is fastest with about six work items per resident slot (4608: within 0.5 % of
// the best; 1536 is 2 % slower).  Smaller grids get fewer (down to 1536: 4 096 x 262 144 takes
// 0.27 ms with 1536 or 4608 items, 0.30 ms with 896), so that the chunk partials stay small.
// "Smaller" is measured in time, not evaluations: `cost` is the model's evaluation time relative to the one-peak
// Lorentzian (sweep_cost<M>).  One rank's 2048 x 524 288 slice of the 7-peak config is as many evaluations as
// 4096 x 262 144 of the one-peak model but runs 1.1 ms: with 1536 items (two rounds of 0.55 ms workgroups) the
// uneven tail costs 5 % — 1.145 / 1.146 ms against 1.093 / 1.083 ms with 4096 (same box, in cycles; 6144 and
// 8192 no better); the one-peak 4096 x 262 144 is indifferent (0.252 vs 0.250-0.254 ms).
static SweepPlan plan_sweep(int64_t ns, int64_t nd, int cost = 1, int max_spt = 8) {
    static const int force_spt = getenv("OBE_SWEEP_SPT") ? atoi(getenv("OBE_SWEEP_SPT")) : 0;         // tuning aids
    static const int force_blocks = getenv("OBE_SWEEP_BLOCKS") ? atoi(getenv("OBE_SWEEP_BLOCKS")) : 0;
    SweepPlan p;
    // (2048 settings: 8 per lane once there are draws enough for the 384 chunks that keep 1536 work items — one
    // rank's 2048 x 524 288 slice of the 10-parameter config: 1.118 vs 1.133 ms, 1.126 vs 1.146 ms, same box)
    p.spt = ns >= 4096 || (ns >= 2048 && nd >= 131072) ? 8 : (ns >= 1024 ? 4 : (ns >= 512 ? 2 : 1));
    if (force_spt == 1 || force_spt == 2 || force_spt == 4 || force_spt == 8) p.spt = force_spt;
    // (a lane keeps 4 running values per setting and channel in registers: models with more than 4 channels get at
    // most 2 settings per lane)
    if (p.spt > max_spt) p.spt = max_spt;
    p.tiles_x = static_cast<int>((ns + (int64_t)kWave * p.spt - 1) / ((int64_t)kWave * p.spt));
    const int64_t by_work = static_cast<int64_t>((double)ns * (double)nd * (double)cost / 1.25e6);
    const int64_t target_blocks = force_blocks > 0 ? force_blocks : std::max<int64_t>(1536, std::min<int64_t>(4608, by_work));
    int64_t want = (target_blocks + p.tiles_x - 1) / p.tiles_x;
    if (want > 8) want = (want + 7) / 8 * 8;          // whole groups of 8 chunks: one per XCD
    const int64_t cap = std::max<int64_t>(1, std::min<int64_t>(kMaxChunks, nd / 256));   // >= 64 draws per wave
    p.nchunks = static_cast<int>(std::max<int64_t>(1, std::min(want, cap)));
    // want and cap never decrease with nd, and rounding the chunk up to whole waves below can only
    // lower the count again: this value bounds the chunks of every sweep of <= nd draws
    p.nchunks_bound = p.nchunks;
    p.chunk = (nd + p.nchunks - 1) / p.nchunks;
    p.chunk = (p.chunk + 63) / 64 * 64;
    p.nchunks = static_cast<int>((nd + p.chunk - 1) / p.chunk);
    return p;
}

static int64_t sweep_ws_doubles(int64_t ns, int64_t nd, int nc, int packed_w) {
    const SweepPlan p = plan_sweep(ns, nd, 1 << 20);       // (the grid of the costliest model: the most chunks)
    // (sized by the monotone bound: the chunk count itself is not monotone in nd after the rounding
    // of the chunk length, and a sweep of N_DRAWS < n_particles draws runs in the same workspace)
    return 2 * (int64_t)p.nchunks_bound * nc * ns  // partial S1, S2
           + (int64_t)nc * ns                      // per-setting shift
           + 3 * argmax_slots(ns) + 16             // argmax / kappa partials + scalars
           + nd * packed_w + 8;                    // packed draws
}

struct SweepArgs {
    obe_model m;
    const double* settings;
    int64_t ld_s, ns;
    const double* particles;
    int64_t ld_p;
    const double* weights;
    const int64_t* draw_idx;   // NULL: all particles, weighted
    int64_t nd;                // draws (== n_particles in full mode)
    int64_t n_particles;
    double uniform_w;          // 1/nd in draws mode
    const double* moments;     // obe_moments output: mean parameters at +2
    int64_t chunk;
    int tiles_x, nchunks;      // logical grid: setting tiles x particle chunks
    int one;                   // 1 (a run-time constant the prefetch address is built from)
    int xcd_map;               // chunks in whole groups of 8: block -> (tile, chunk) follows the XCD round robin
    double* packed;            // (nd, packed_width): written by sweep_pack_kernel, read by sweep_kernel
    double* part1;
    double* part2;
    double* cs_out;            // (C, ns): the shift each setting used (0 when unshifted)
    const unsigned* abort;     // OBE_SWEEP_SPECULATIVE: the workspace's abort word (non-zero: do nothing), else NULL
};

// A speculative sweep (enqueued behind an update whose resample decision was not waited for) does nothing
// when that update said "resample": every kernel of the call starts with this test.
__device__ __forceinline__ bool sweep_aborted(const unsigned* abort) {
    return abort && __builtin_amdgcn_readfirstlane(*abort) != 0u;
}

// Every draw packed once per sweep: M::pack() (per-particle divisions, sqrt(w) folded into the
// amplitudes) and sqrt(w), one 16-byte-aligned record per draw.
template <class M>
__global__ __launch_bounds__(kBlock) void sweep_pack_kernel(SweepArgs a) {
    constexpr int NPK = M::NPK, NPKW = packed_width<M>();
    // records wider than 32 bytes leave through LDS (round 5): a thread storing its own 80-byte record makes every
    // store instruction of a wave touch 40 cache lines, 16 bytes each; staged, a workgroup's 256 records are one
    // contiguous run written lane by lane
    constexpr bool STAGED = NPKW > 4 && NPKW <= 24;          // (<= 48 KB of LDS)
    __shared__ double2 tile[STAGED ? kBlock * NPKW / 2 : 1];
    if (sweep_aborted(a.abort)) return;
    const double* __restrict__ thbar = a.moments + 2;   // weighted-mean parameters (K3 output)
    for (int64_t p0 = (int64_t)blockIdx.x * kBlock; p0 < a.nd; p0 += (int64_t)gridDim.x * kBlock) {
        const int64_t p = p0 + threadIdx.x;
        double pk[NPKW];
        if (p < a.nd) {
            int64_t src = p;
            double w;
            if (a.draw_idx) {
                src = a.draw_idx[p];
                src = src < 0 ? 0 : (src >= a.n_particles ? a.n_particles - 1 : src);
                w = a.uniform_w;
            } else {
                w = a.weights[p];
            }
            const double sw = sqrt(w);
            M::pack(ParamRef{a.particles + src, a.ld_p}, thbar, a.m, sw, pk);
            pk[NPK] = sw;
#pragma unroll
            for (int k = NPK + 1; k < NPKW; ++k) pk[k] = 0.0;
        }
        if constexpr (STAGED) {
            __syncthreads();           // the previous trip's tile has left
            if (p < a.nd) {
#pragma unroll
                for (int k = 0; k < NPKW / 2; ++k) tile[threadIdx.x * (NPKW / 2) + k] = double2{pk[2 * k], pk[2 * k + 1]};
            }
            __syncthreads();
            const int64_t run = (a.nd - p0 < kBlock ? a.nd - p0 : kBlock) * (NPKW / 2);
            double2* __restrict__ out = reinterpret_cast<double2*>(a.packed + p0 * NPKW);
            for (int64_t e = threadIdx.x; e < run; e += kBlock) out[e] = tile[e];
        } else if (p < a.nd) {
            double2* __restrict__ out = reinterpret_cast<double2*>(a.packed + p * NPKW);
#pragma unroll
            for (int k = 0; k < NPKW / 2; ++k) out[k] = double2{pk[2 * k], pk[2 * k + 1]};
        }
    }
}

// a wave-uniform 64-bit value, held in scalar registers
__device__ __forceinline__ int64_t wave_uniform(int64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
    const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(static_cast<uint64_t>(v) >> 32));
    return static_cast<int64_t>((static_cast<uint64_t>(hi) << 32) | lo);
}

// SAFE (models with kHasSafeEval only): evaluate with the model's sweep_eval_safe() — the repeat
// after a sweep whose fast, branch-free batch inversions poisoned a variance (kappa = NaN).
#ifdef OBE_SWEEP_WAVES_PER_EU          // tuning aid (tools/build_variant.py): force an occupancy
#define OBE_SWEEP_OCCUPANCY __attribute__((amdgpu_waves_per_eu(OBE_SWEEP_WAVES_PER_EU, OBE_SWEEP_WAVES_PER_EU)))
#else
#define OBE_SWEEP_OCCUPANCY
#endif
template <class M, int SPT, bool SHIFT, bool SAFE = false>
__global__ __launch_bounds__(kBlock) OBE_SWEEP_OCCUPANCY void sweep_kernel(SweepArgs a) {
    constexpr int NC = M::NC, NXS = M::NXS, NPK = M::NPK, NPKW = packed_width<M>();
    constexpr bool PAIRS = SPT >= 2 && (SAFE ? safe_pair_eval<M>::value : has_pair_eval<M>::value);
    // particles per prefetched group: two groups of packed particles live in SGPRs (~100 per wave)
    constexpr int G = PAIRS ? (NPKW <= 4 ? 4 : 2) : (NPKW <= 4 ? 4 : (NPKW <= 8 ? 2 : 1));
    __shared__ double red[kSweepWaves][NC][2][kWave];

    // XCD-aware block -> (setting tile, particle chunk) map.  Workgroup b is dispatched to
    // XCD b % 8 (observed placement; only speed depends on it): give XCD x the chunks
    // {x, x+8, ...}, so each 4 MiB L2 streams 1/8 of the cloud instead of all of it
    // (rocprofv3 FETCH_SIZE before: 8 x the cloud per launch).
    // (Only when the chunks come in whole groups of 8; a sweep of few draws has fewer chunks than
    // XCDs, and padding the map would leave most of its workgroups — and XCDs — without work.)
    int chunk_id, tile_x;
    if (a.xcd_map) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        chunk_id = (slot / a.tiles_x) * 8 + xcd;
        tile_x = slot % a.tiles_x;
    } else {
        chunk_id = blockIdx.x / a.tiles_x;
        tile_x = blockIdx.x % a.tiles_x;
    }
    if (chunk_id >= a.nchunks || sweep_aborted(a.abort)) return;
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;

    double xs[SPT][NXS], cs[SPT][NC], s1[SPT][NC], s2[SPT][NC];
    const double* __restrict__ thbar = a.moments + 2;   // weighted-mean parameters (K3 output)

    {   // prepared settings and the per-setting shift c_s = y'(x_s; mean parameters)
        double pkbar[NPK];
        M::pack(ParamRef{thbar, 1}, thbar, a.m, 1.0, pkbar);
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            int64_t s = ((int64_t)tile_x * SPT + j) * kWave + lane;
            if (s >= a.ns) s = a.ns - 1;
            double x[M::NS];
#pragma unroll
            for (int k = 0; k < M::NS; ++k) x[k] = a.settings[(int64_t)k * a.ld_s + s];
            M::prep_setting(x, a.m, xs[j]);
#pragma unroll
            for (int c = 0; c < NC; ++c) s1[j][c] = s2[j][c] = 0.0;
        }
        if constexpr (SAFE) M::template sweep_eval_safe<SPT>(xs, pkbar, 1.0, a.m, cs);
        else M::template sweep_eval<SPT>(xs, pkbar, 1.0, a.m, cs);
        if (!SHIFT) {
#pragma unroll
            for (int j = 0; j < SPT; ++j)
#pragma unroll
                for (int c = 0; c < NC; ++c) cs[j][c] = 0.0;
        }
    }

    // this wave's quarter of the chunk
    const int64_t c_begin = (int64_t)chunk_id * a.chunk;
    const int64_t c_end = c_begin + a.chunk < a.nd ? c_begin + a.chunk : a.nd;
    const int64_t per = ((c_end - c_begin + 4 * kSweepWaves - 1) / (4 * kSweepWaves)) * 4;
    int64_t p_begin = c_begin + wid * per;
    if (p_begin > c_end) p_begin = c_end;
    const int64_t p_end = p_begin + per < c_end ? p_begin + per : c_end;
    const int n = static_cast<int>(wave_uniform(p_end - p_begin));
    const double* __restrict__ pk = a.packed + wave_uniform(p_begin * NPKW);   // uniform address: scalar loads

    auto accumulate = [&](const double (&v)[SPT][NC], double sw) {
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const double u = SHIFT ? fma(-cs[j][c], sw, v[j][c]) : v[j][c];   // sqrt(w) * (y' - c_s)
                s1[j][c] = fma(sw, u, s1[j][c]);                     // sum w (y' - c_s)
                s2[j][c] = fma(u, u, s2[j][c]);                      // sum w (y' - c_s)^2
            }
        }
    };
    auto load_group = [&](int i0, double (&g)[G][NPKW]) {
#pragma unroll
        for (int e = 0; e < G; ++e)
#pragma unroll
            for (int k = 0; k < NPKW; ++k) g[e][k] = pk[(i0 + e) * NPKW + k];
    };
    auto process_group = [&](const double (&g)[G][NPKW]) {
        if constexpr (PAIRS) {
#pragma unroll
            for (int e = 0; e < G; e += 2) {              // two particles share one reciprocal
                double va[SPT][NC], vb[SPT][NC];
                if constexpr (SAFE) M::template sweep_eval_pair<SPT, true>(xs, g[e], g[e + 1], va, vb);
                else M::template sweep_eval_pair<SPT>(xs, g[e], g[e + 1], va, vb);
                accumulate(va, g[e][NPK]);
                accumulate(vb, g[e + 1][NPK]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < G; ++e) {
                double v[SPT][NC];
                if constexpr (SAFE) M::template sweep_eval_safe<SPT>(xs, g[e], g[e][NPK], a.m, v);
                else M::template sweep_eval<SPT>(xs, g[e], g[e][NPK], a.m, v);   // sqrt(w) * y'
                accumulate(v, g[e][NPK]);
            }
        }
    };

    int i = 0;
    if (n >= G) {
        // software pipeline: the next group's scalar loads are in flight while this one is evaluated
        // (an L2 round trip is ~1 us; a group of the Lorentzian is 4 x 8 x 7.4 issue slots ~ 0.4 us)
        double cur[G][NPKW];
        load_group(0, cur);
        asm("" : "+s"(cur[0][0]));     // the first group has landed before the loop: no wait at the loop head,
                                       // where it would also wait for the prefetch just issued
        for (; i + G <= n; i += G) {
            double nxt[G][NPKW];
            // (a.one == 1, unknown to the optimizer: it must not fold the prefetch into the next trip's
            // own load; the last trip re-reads its own group)
            load_group(i + 2 * G <= n ? i + G * a.one : i, nxt);
            __builtin_amdgcn_sched_barrier(0);                // the loads stay ahead of the arithmetic
            process_group(cur);
#pragma unroll
            for (int e = 0; e < G; ++e)
#pragma unroll
                for (int k = 0; k < NPKW; ++k) cur[e][k] = nxt[e][k];
        }
    }
    for (; i < n; ++i) {
        double g[NPKW], v[SPT][NC];
#pragma unroll
        for (int k = 0; k < NPKW; ++k) g[k] = pk[i * NPKW + k];
        if constexpr (SAFE) M::template sweep_eval_safe<SPT>(xs, g, g[NPK], a.m, v);
        else M::template sweep_eval<SPT>(xs, g, g[NPK], a.m, v);
        accumulate(v, g[NPK]);
    }

    // (global stores only after the streaming loop: nothing may alias the packed draws before it,
    // or their loads would not be scalar)
    if (chunk_id == 0 && wid == 0) {
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const int64_t s = ((int64_t)tile_x * SPT + j) * kWave + lane;
            if (s < a.ns) {
#pragma unroll
                for (int c = 0; c < NC; ++c) a.cs_out[(int64_t)c * a.ns + s] = cs[j][c];
            }
        }
    }
    // the four waves' moments, added in wave order (fixed association); wave j % 4 writes setting j
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            red[wid][c][0][lane] = s1[j][c];
            red[wid][c][1][lane] = s2[j][c];
        }
        __syncthreads();
        const int64_t s = ((int64_t)tile_x * SPT + j) * kWave + lane;
        if (wid == (j & (kSweepWaves - 1)) && s < a.ns) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                double t1 = red[0][c][0][lane], t2 = red[0][c][1][lane];
#pragma unroll
                for (int g = 1; g < kSweepWaves; ++g) {
                    t1 += red[g][c][0][lane];
                    t2 += red[g][c][1][lane];
                }
                const int64_t o = ((int64_t)chunk_id * NC + c) * a.ns + s;
                a.part1[o] = t1;
                a.part2[o] = t2;
            }
        }
    }
}

struct UtilArgs {
    const double* noise_var;
    int64_t noise_ld;     // 0: one value per channel; > 0: (C, n_settings) rows this far apart; < 0: see make_util_args
    const double* cost;   // NULL: scalar
    double cost_scalar;
    int mom_dims;         // noise_ld < 0: noise_var is a K3 block of this many parameters ...
    int mom_rows[OBE_MAX_CHANNELS];   // ... and channel c's noise variance is its m2[row] / sum w
};

// noise_ld < 0 (OBE_NOISE_FROM_MOMENTS): the noise variance of channel c is np.average(sigma_c^2, weights=w)
// (obe_noiseparam.py:122-136) = m2[row_c] / sum w of the K3 block d_noise_var points to — the division that
// obe_noise_var_from_moments()'s kernel does, done by the kernel that needs the value: one launch (4-5 us between
// the update and every sweep of a noise-parameter object) less, the same bits.
static int make_util_args(UtilArgs& ua, const double* d_noise_var, int64_t noise_ld, const double* d_cost, double cost_scalar,
                          int n_channels, int n_params) {
    ua = UtilArgs{d_noise_var, noise_ld, d_cost, cost_scalar, 0, {}};
    if (noise_ld >= 0) return 0;
    const int64_t code = -noise_ld - 1;
    ua.mom_dims = n_params;
    for (int c = 0; c < OBE_MAX_CHANNELS; ++c) {
        ua.mom_rows[c] = (int)((code >> (5 * c)) & 31);
        if (c < n_channels && ua.mom_rows[c] >= n_params) return bad_arg("noise rows encoded in noise_ld: out of range");
    }
    if (n_params < 1 || n_params > OBE_MAX_DIMS) return bad_arg("noise_ld < 0 needs the number of parameters");
    return 0;
}

__device__ __forceinline__ double utility_of(const double* var, int nc, int64_t s, const UtilArgs& u) {
    // np.sum(var_p / var_n, axis=0) / cost   (obe_base.py:654-655)
    double acc = 0.0;
    for (int c = 0; c < nc; ++c) {
        const double nv = u.noise_ld > 0 ? u.noise_var[(int64_t)c * u.noise_ld + s]
                          : (u.noise_ld == 0 ? u.noise_var[c]
                                             : u.noise_var[2 + 2 * u.mom_dims + u.mom_rows[c]] / u.noise_var[0]);
        acc = acc + var[c] / nv;
    }
    return acc / (u.cost ? u.cost[s] : u.cost_scalar);
}

__device__ __forceinline__ void block_argmax(Best b, double* bv, int64_t* bi) {
    __shared__ double sv[kBlock];
    __shared__ int64_t si[kBlock];
    sv[threadIdx.x] = b.v;
    si[threadIdx.x] = b.i;
    __syncthreads();
    for (int o = kBlock / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            Best x{sv[threadIdx.x], si[threadIdx.x]}, y{sv[threadIdx.x + o], si[threadIdx.x + o]};
            if (better(y, x)) {
                sv[threadIdx.x] = y.v;
                si[threadIdx.x] = y.i;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        bv[blockIdx.x] = sv[0];
        bi[blockIdx.x] = si[0];
    }
}

// folds the chunk partials -> yvar, utility; per-workgroup first maximum and worst kappa (argmax_fold finishes).
// A workgroup owns 64 consecutive settings; its FG wavefronts each sum an FG-th of the chunks (coalesced
// 512-byte rows, up to 8 independent loads in flight per lane), the groups are combined in a fixed order
// through LDS.  (One thread per setting walking all chunks serially was latency-bound: 100+ us at 512
// chunks.)  FG = 16 when there are many chunks per setting — a 4096 x 262 144 sweep has 192 of them and only
// 64 finalize workgroups: 4 wavefronts with 4 loads in flight took 10.8 us for its 12.6 MB, 48 dependent
// trips per wave, 16 wavefronts take 7.9 us — FG = 4 when there are few (65 536 x 1 048 576: 40 chunks,
// 1024 workgroups, bandwidth-bound: 11.3 us with 4, 14.3 us with 16).
// Round 4, measured and not kept: argmax_fold's work done here by the last workgroup to arrive (write-through
// entries, arrival counter).  With 64-1024 workgroups the serialised tickets (~12 ns each) and the tail of
// the last workgroup cost more than the launch they save: 14.3 vs 7.9 + 4.9 us (4096 x 262 144), 27.0 vs
// 14.3 + 7.0 us (65 536 x 1 048 576), 18.1 vs 9.6 + 5.2 us (16 384 x 524 288, 10 parameters).  With the two-level
// tickets of obe_common.h: 14.6 vs 8.1 + 5.2, 18.2 vs 10.6 + 7.9, 16.4 vs 9.7 + 5.9 us — level with the two
// launches and their 1.7 us boundary, no better: the last workgroup's tail (drain, tickets, reading the
// entries past L1, the host words) costs what argmax_fold costs.
constexpr int kFinGroupsMany = 16, kFinGroupsFew = 4;      // chunk groups = wavefronts of a finalize workgroup
constexpr int kFinManyChunks = 64;
constexpr int kFinNarrow = 16;             // settings per workgroup of the narrow form (see sweep_finalize) ...
constexpr int kFinNarrowBelow = 128;       // ... used while 64 settings per workgroup would make fewer workgroups than this

// worst cancellation factor so far; a NaN (some variance is NaN) is sticky
__device__ __forceinline__ double kappa_worst(double a, double b) {
    return a != a ? a : (b != b ? b : (a > b ? a : b));
}

// Where the result goes on the host, as device-visible addresses (page-locked host memory), or all
// NULL: then read_best() copies it.
struct HostResult {
    double* best;
    int64_t* idx;
    double* kappa;
    double* tail = nullptr;     // the result record once more, at the END of the workspace (OBE_WS_RESULT_TAIL), or NULL
};
// (every word is armed by the host and waited for on its own: the order of the stores does not matter)
__device__ __forceinline__ void deliver(const HostResult& h, double v, int64_t i, double k) {
    if (h.tail) {       // where the update calls that reuse the head of the workspace do not reach
        h.tail[0] = v;
        reinterpret_cast<int64_t*>(h.tail)[1] = i;
        h.tail[2] = k;
        h.tail[3] = 0.0;
    }
    if (h.idx) {
        if (h.best) *h.best = v;
        if (h.kappa) *h.kappa = k;
        host_results_before_flag();
        *h.idx = i;
    } else if (h.kappa) {
        if (h.best) *h.best = v;
        host_results_before_flag();
        *h.kappa = k;
    } else if (h.best) {
        *h.best = v;
    }
}

// first maximum (np.argmax order) and worst kappa over a wavefront; valid in lane 0
__device__ __forceinline__ void wave_best(Best& b, double& kappa) {
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        const Best y{__shfl_down(b.v, o, kWave), __shfl_down(b.i, o, kWave)};
        if (better(y, b)) b = y;
        kappa = kappa_worst(kappa, __shfl_down(kappa, o, kWave));
    }
}

// the result record of a sweep / argmax call: out_v[0] best value, [1] kappa, out_i[0] index, and the same as
// one contiguous 32-byte record {value, index bits, kappa, 0} at OBE_WS_RESULT_OFFSET (out_v + 2), so that a
// sharded caller can all-gather it straight from device memory without a host round trip
__device__ __forceinline__ void write_result_record(double* out_v, int64_t* out_i, const Best& b, double k,
                                                    const HostResult& host) {
    out_v[0] = b.v;
    out_i[0] = b.i;
    out_v[1] = k;
    out_v[2] = b.v;
    reinterpret_cast<int64_t*>(out_v)[3] = b.i;
    out_v[4] = k;
    out_v[5] = 0.0;
    deliver(host, b.v, b.i, k);
}

// WS = settings per workgroup: 64 (a wavefront per chunk group), or 16 — then a wavefront holds FOUR chunk groups
// of 16 settings each and the workgroup is a quarter as large, so that a grid of few settings still spreads over
// the chip: 2048 settings x 1024 chunks (one rank's slice of the 7-peak config, 33.5 MB of partials) was 32
// workgroups = 32 CUs, 20.4 us; 128 workgroups: see DESIGN.md K1.  Every setting's chunks are summed by the same
// FG groups in the same order either way: identical bits.
template <int FG, int WS = kFinSettings>
__global__ __launch_bounds__(FG * WS) void sweep_finalize(const double* __restrict__ part1,
                                                             const double* __restrict__ part2, int nchunks, int nc,
                                                             int64_t ns, const double* __restrict__ moments,
                                                             int full_mode, UtilArgs ua,
                                                             const double* __restrict__ cs,
                                                             double* __restrict__ yvar,
                                                             double* __restrict__ utility, double* __restrict__ bv,
                                                             int64_t* __restrict__ bi, double* __restrict__ bk,
                                                             const unsigned* abort) {
    static_assert(WS == 64 || WS == 16, "a wavefront holds one or four chunk groups");
    // (one channel's group sums at a time: OBE_MAX_CHANNELS = 8 of them would not fit the 64 KB of static LDS at
    // FG = 16, WS = 64.  Wavefront 0 turns channel c's sums into its variance before channel c + 1 overwrites them;
    // the sums themselves, their order and the arithmetic on them are what they were: the same bits.)
    __shared__ double acc1[FG][WS];
    __shared__ double acc2[FG][WS];
    if (sweep_aborted(abort)) return;
    const double W = full_mode ? moments[0] : 1.0;
    const int wlane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int lane = wlane % WS, grp = wave * (kWave / WS) + wlane / WS;      // setting within the tile, chunk group
    const int64_t s = (int64_t)blockIdx.x * WS + lane;
    double var[OBE_MAX_CHANNELS];
    double kappa = 0.0;     // worst (mean of y')^2 / var: the cancellation an UNSHIFTED sweep would suffer
    for (int c = 0; c < nc; ++c) {
        double a1 = 0.0, a2 = 0.0;
        if (s < ns) {
            int k = grp;
            // (8 loads in flight.  Round 5, measured and not kept: 16 per trip — the compiler splits them 5 + 11 with a
            // full wait in between, two round trips as before; a rank's 2048 x 1024 chunks: 12.0 us against 11.2)
            for (; k + 3 * FG < nchunks; k += 4 * FG) {
                double t1[4], t2[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t o = ((int64_t)(k + u * FG) * nc + c) * ns + s;
                    t1[u] = part1[o];
                    t2[u] = part2[o];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    a1 += t1[u];
                    a2 += t2[u];
                }
            }
            if (k < nchunks) {          // the last 1-3 chunks of this group: one batch (clamped, added selectively)
                double t1[3], t2[3];
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int kk = k + u * FG;
                    const int64_t o = ((int64_t)(kk < nchunks ? kk : k) * nc + c) * ns + s;
                    t1[u] = part1[o];
                    t2[u] = part2[o];
                }
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const double n1 = a1 + t1[u], n2 = a2 + t2[u];
                    const bool ok = k + u * FG < nchunks;
                    a1 = ok ? n1 : a1;
                    a2 = ok ? n2 : a2;
                }
            }
        }
        if (c > 0) __syncthreads();          // wavefront 0 has consumed the previous channel's sums
        acc1[grp][lane] = a1;
        acc2[grp][lane] = a2;
        __syncthreads();
        if (wave == 0 && grp == 0 && s < ns) {
            a1 = 0.0;
            a2 = 0.0;
#pragma unroll
            for (int g = 0; g < FG; ++g) {
                a1 += acc1[g][lane];
                a2 += acc2[g][lane];
            }
            // (Measured alternative: chunk partials combined with TwoSum and S1*(S1/W) formed exactly
            // with FMAs, plus per-tile flushing of the running sums, lowers the error of the
            // unshifted variance from ~1e-15*kappa to ~2e-16*kappa — but the exact product then
            // exposes the rounding of S2 itself, e.g. a non-zero variance for a single draw where
            // the plain formula cancels to the reference's exact 0.  Not kept.)
            const double mu = a1 / W;
            double v = (a2 - a1 * mu) / W;
            v = v > 0.0 ? v : (v != v ? v : 0.0);          // rounding may leave -0 / tiny negatives; NaN stays NaN (np.var)
            var[c] = v;
            yvar[(int64_t)c * ns + s] = v;
            const double m = cs[(int64_t)c * ns + s] + mu;
            const double k = v != v ? v : (v > 0.0 ? (m * m) / v : (m == 0.0 ? 0.0 : INFINITY));
            kappa = kappa_worst(kappa, k);                 // a NaN variance is reported as kappa = NaN
        }
    }
    if (wave != 0) return;
    Best best{-INFINITY, INT64_MAX};
    if (grp == 0 && s < ns) {
        const double u = utility_of(var, nc, s, ua);
        utility[s] = u;
        best = Best{u, s};
    }
    wave_best(best, kappa);
    if (wlane == 0) {
        bv[blockIdx.x] = best.v;
        bi[blockIdx.x] = best.i;
        bk[blockIdx.x] = kappa;
    }
}

__global__ __launch_bounds__(kBlock) void utility_kernel(const double* __restrict__ yvar, int nc, int64_t ns,
                                                         UtilArgs ua, double* __restrict__ utility,
                                                         double* __restrict__ bv, int64_t* __restrict__ bi) {
    Best best{-INFINITY, INT64_MAX};
    for (int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x; s < ns; s += (int64_t)gridDim.x * kBlock) {
        // np.sum(var_p / var_n, axis=0) / cost over ANY number of channels (obe_base.py:654-655; noise_ld >= 0 here)
        double acc = 0.0;
        for (int c = 0; c < nc; ++c) {
            const double nv = ua.noise_ld > 0 ? ua.noise_var[(int64_t)c * ua.noise_ld + s] : ua.noise_var[c];
            acc = acc + yvar[(int64_t)c * ns + s] / nv;
        }
        const double u = acc / (ua.cost ? ua.cost[s] : ua.cost_scalar);
        utility[s] = u;
        Best cand{u, s};
        if (better(cand, best)) best = cand;
    }
    block_argmax(best, bv, bi);
}

__global__ __launch_bounds__(kBlock) void argmax_kernel(const double* __restrict__ v, int64_t n,
                                                        double* __restrict__ bv, int64_t* __restrict__ bi) {
    Best best{-INFINITY, INT64_MAX};
    for (int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x; s < n; s += (int64_t)gridDim.x * kBlock) {
        Best cand{v[s], s};
        if (better(cand, best)) best = cand;
    }
    block_argmax(best, bv, bi);
}

// one block: first-max over the block partials -> scalars {value, index (as int64 bits)}
__global__ __launch_bounds__(kBlock) void argmax_fold(const double* __restrict__ bv, const int64_t* __restrict__ bi,
                                                      int nb, const double* __restrict__ bk,
                                                      double* __restrict__ out_v,
                                                      int64_t* __restrict__ out_i, HostResult host,
                                                      const unsigned* abort = nullptr) {
    if (sweep_aborted(abort)) return;       // (the armed host words stay armed: nobody reads this result)
    Best best{-INFINITY, INT64_MAX};
    double kmax = 0.0;                  // worst cancellation factor; NaN is sticky
    // (the partials of four rounds of threads — value, index, kappa: up to 12 loads — in flight together; compared in
    // the order b, b + 256, ... of the one-at-a-time loops: the same winner.  1024 partials used to be 8 dependent
    // round trips in a one-workgroup kernel whose whole duration is latency)
    for (int base = 0; base < nb; base += 4 * kBlock) {
        double v[4], kk[4];
        int64_t ix[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = base + r * kBlock + (int)threadIdx.x, bc = b < nb ? b : nb - 1;
            v[r] = bv[bc];
            ix[r] = bi[bc];
            kk[r] = bk ? bk[bc] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = base + r * kBlock + (int)threadIdx.x;
            if (b < nb) {
                const Best cand{v[r], ix[r]};
                if (better(cand, best)) best = cand;
                kmax = kappa_worst(kmax, kk[r]);
            }
        }
    }
    block_argmax(best, out_v, out_i);   // gridDim.x == 1 -> writes element 0
    __shared__ double kred[kBlock];
    kred[threadIdx.x] = kmax;
    __syncthreads();
    for (int o = kBlock / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) kred[threadIdx.x] = kappa_worst(kred[threadIdx.x], kred[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) write_result_record(out_v, out_i, Best{out_v[0], out_i[0]}, kred[0], host);
}

// Reference-semantics sweeps are tiny (201 settings x 30 draws in the demos): one workgroup
// does what sweep_kernel + sweep_finalize + argmax_fold do in three launches — the draws are
// packed into LDS once, each thread walks its settings, and the block folds utility, first
// maximum and kappa.  Same formulas and summation order as the SPT = 1, one-chunk path of the
// big kernels, hence the same bits.
template <class M, bool SAFE>
__global__ __launch_bounds__(kBlock) void sweep_small_kernel(SweepArgs a, UtilArgs ua, double* __restrict__ yvar,
                                                             double* __restrict__ utility,
                                                             double* __restrict__ out_v,
                                                             int64_t* __restrict__ out_i, HostResult host) {
    constexpr int NC = M::NC, NXS = M::NXS, NPK = M::NPK;
    constexpr int NPKW = (NPK + 1 + 1) & ~1;
    extern __shared__ __attribute__((aligned(16))) double tile[];
    __shared__ double kred[kBlock];
    const double* __restrict__ thbar = a.moments + 2;
    const int nd = static_cast<int>(a.nd);
    for (int i = threadIdx.x; i < nd; i += kBlock) {
        int64_t src = a.draw_idx[i];
        src = src < 0 ? 0 : (src >= a.n_particles ? a.n_particles - 1 : src);
        const double sw = sqrt(a.uniform_w);
        double pk[NPK];
        M::pack(ParamRef{a.particles + src, a.ld_p}, thbar, a.m, sw, pk);
#pragma unroll
        for (int k = 0; k < NPK; ++k) tile[i * NPKW + k] = pk[k];
        tile[i * NPKW + NPK] = sw;
    }
    __syncthreads();
    double pkbar[NPK];
    M::pack(ParamRef{thbar, 1}, thbar, a.m, 1.0, pkbar);
    auto eval = [&](const double (&xs)[1][NXS], const double* pk, double sw, double (&v)[1][NC]) {
        if constexpr (SAFE) M::template sweep_eval_safe<1>(xs, pk, sw, a.m, v);
        else M::template sweep_eval<1>(xs, pk, sw, a.m, v);
    };
    const double W = 1.0;      // draws mode: uniform weights 1/nd
    Best best{-INFINITY, INT64_MAX};
    double kappa = 0.0;
    for (int64_t s = threadIdx.x; s < a.ns; s += kBlock) {
        double x[M::NS], xs[1][NXS], cs[1][NC], s1[NC], s2[NC];
#pragma unroll
        for (int k = 0; k < M::NS; ++k) x[k] = a.settings[(int64_t)k * a.ld_s + s];
        M::prep_setting(x, a.m, xs[0]);
        eval(xs, pkbar, 1.0, cs);
#pragma unroll
        for (int c = 0; c < NC; ++c) s1[c] = s2[c] = 0.0;
        for (int i = 0; i < nd; ++i) {
            double pk[NPK], v[1][NC];
#pragma unroll
            for (int k = 0; k < NPK; ++k) pk[k] = tile[i * NPKW + k];
            const double sw = tile[i * NPKW + NPK];
            eval(xs, pk, sw, v);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const double u = fma(-cs[0][c], sw, v[0][c]);
                s1[c] = fma(sw, u, s1[c]);
                s2[c] = fma(u, u, s2[c]);
            }
        }
        double var[OBE_MAX_CHANNELS];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const double mu = s1[c] / W;
            double v = (s2[c] - s1[c] * mu) / W;
            v = v > 0.0 ? v : (v != v ? v : 0.0);
            var[c] = v;
            yvar[(int64_t)c * a.ns + s] = v;
            const double m = cs[0][c] + mu;
            const double k = v != v ? v : (v > 0.0 ? (m * m) / v : (m == 0.0 ? 0.0 : INFINITY));
            kappa = kappa_worst(kappa, k);
        }
        const double u = utility_of(var, NC, s, ua);
        utility[s] = u;
        const Best cand{u, s};
        if (better(cand, best)) best = cand;
    }
    block_argmax(best, out_v, out_i);       // one block: element 0
    kred[threadIdx.x] = kappa;
    __syncthreads();
    for (int o = kBlock / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) kred[threadIdx.x] = kappa_worst(kred[threadIdx.x], kred[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {                 // the result record, as argmax_fold leaves it
        const double k = kred[0];
        out_v[1] = k;
        out_v[2] = out_v[0];
        reinterpret_cast<int64_t*>(out_v)[3] = out_i[0];
        out_v[4] = k;
        out_v[5] = 0.0;
        deliver(host, out_v[0], out_i[0], k);
    }
}

// np.var(utility_y_space, axis=0): two-pass over the (small) draw axis
__global__ __launch_bounds__(kBlock) void yspace_var_kernel(const double* __restrict__ ysp, int64_t nd,
                                                            int64_t row /* C*Ns */, double* __restrict__ yvar) {
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < row; e += (int64_t)gridDim.x * kBlock) {
        // (eight draws' loads in flight together, added in draw order as before: one load / wait / add per draw
        // made the 30 draws of a reference-semantics cycle 60 dependent round trips)
        double sum = 0.0;
        for (int64_t d0 = 0; d0 < nd; d0 += 8) {
            double t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = ysp[(d0 + j < nd ? d0 + j : nd - 1) * row + e];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const double n = sum + t[j];
                sum = d0 + j < nd ? n : sum;
            }
        }
        const double mean = sum / (double)nd;
        double acc = 0.0;
        for (int64_t d0 = 0; d0 < nd; d0 += 8) {
            double t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = ysp[(d0 + j < nd ? d0 + j : nd - 1) * row + e];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const double dv = t[j] - mean;
                const double sq = dv * dv;
                const double n = acc + sq;
                acc = d0 + j < nd ? n : acc;
            }
        }
        yvar[e] = acc / (double)nd;
    }
}

// Optional timing of the sweep kernel inside real cycles (obe_sweep_timing): events around the
// launch on its own stream, read after the stream synchronisation the result copy needs anyway.
struct SweepTiming {
    bool on = false;
    int device = -1;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    double total_ms = 0.0;
    int64_t launches = 0;
    bool pending = false;      // a speculative call's events have not been read yet
    double pending_min_ms = 0.0;
};
static SweepTiming g_timing;

// A speculative call returns before its kernels have run: its pair of events is read at the next timed call
// or query.  An aborted sweep (its update resampled) is not a sweep: shorter than any kernel could sweep
// that many evaluations (half of 8e12 evaluations/s, well above this chip's rate), it is left out.
static void resolve_pending_timing() {
    if (!g_timing.pending) return;
    g_timing.pending = false;
    float ms = 0.f;
    if (hipEventSynchronize(g_timing.e1) != hipSuccess) return;
    if (hipEventElapsedTime(&ms, g_timing.e0, g_timing.e1) != hipSuccess) return;
    if (ms >= g_timing.pending_min_ms) {
        g_timing.total_ms += ms;
        g_timing.launches += 1;
    }
}

struct SweepWs {
    double* part1;
    double* part2;
    double* bv;
    int64_t* bi;
    double* out_v;
    int64_t* out_i;
    double* bk;
    double* cs;
    double* packed;
};

static int64_t sweep_ws_bytes(int64_t part_doubles, int64_t cs_doubles, int64_t slots, int64_t packed_doubles) {
    return (2 * part_doubles + cs_doubles + 3 * slots + 16 + packed_doubles + 2) * (int64_t)sizeof(double);
}
static int carve_sweep_ws(void* d_ws, int64_t ws_bytes, int64_t part_doubles, int64_t cs_doubles, SweepWs& w,
                          int64_t slots = kMaxBlocks, int64_t packed_doubles = 0) {
    const int64_t need = sweep_ws_bytes(part_doubles, cs_doubles, slots, packed_doubles);
    if (!d_ws || ws_bytes < need) return bad_arg("sweep workspace too small");
    double* base = static_cast<double*>(d_ws);
    w.out_v = base;                                     // [0] best value, [1] worst cancellation factor
    w.out_i = reinterpret_cast<int64_t*>(base + 8);     // [8]
    w.bv = base + 16;
    w.bi = reinterpret_cast<int64_t*>(base + 16 + slots);
    w.bk = base + 16 + 2 * slots;
    w.cs = base + 16 + 3 * slots;
    w.part1 = w.cs + cs_doubles;
    w.part2 = w.part1 + part_doubles;
    double* pk = w.part2 + part_doubles;                // 16-byte aligned records
    w.packed = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(pk) + 15) & ~uintptr_t(15));
    return 0;
}

// the copy of the result record at the end of the workspace (include/obe_hip.h: OBE_WS_RESULT_TAIL), if the
// workspace has the 48 spare bytes behind what the call itself uses
static double* result_tail(void* d_ws, int64_t ws_bytes, int64_t need_bytes) {
    if (!d_ws || ws_bytes < need_bytes + 48) return nullptr;
    return reinterpret_cast<double*>(static_cast<char*>(d_ws) + (ws_bytes & ~(int64_t)7) - 48);
}

// the device views of the caller's host outputs if every one that is asked for is page-locked; each such
// word is armed (before any launch) and read_best() waits for each of them
static HostResult host_result(double* h_best, int64_t* h_best_idx, double* h_kappa) {
    HostResult r{static_cast<double*>(device_view_of_host(h_best)), static_cast<int64_t*>(device_view_of_host(h_best_idx)),
                 static_cast<double*>(device_view_of_host(h_kappa))};
    if ((h_best && !r.best) || (h_best_idx && !r.idx) || (h_kappa && !r.kappa)) r = HostResult{nullptr, nullptr, nullptr};
    if (r.best) arm_host_word(h_best);
    if (r.idx) arm_host_word(h_best_idx);
    if (r.kappa) arm_host_word(h_kappa);
    return r;
}

static int read_best(const SweepWs& w, double* h_best, int64_t* h_best_idx, hipStream_t st,
                     double* h_kappa = nullptr, const HostResult& delivered = HostResult{nullptr, nullptr, nullptr}) {
    if (!h_best && !h_best_idx && !h_kappa) return 0;
    if (delivered.best || delivered.idx || delivered.kappa) {       // the kernel writes them: watch every word
        if (h_best_idx) if (int rc = wait_host_word(h_best_idx, st)) return rc;
        if (h_kappa) if (int rc = wait_host_word(h_kappa, st)) return rc;
        if (h_best) if (int rc = wait_host_word(h_best, st)) return rc;
        return 0;
    }
    double tmp[9];
    OBE_HIP_TRY(hipMemcpyAsync(tmp, w.out_v, 9 * sizeof(double), hipMemcpyDeviceToHost, st));
    OBE_HIP_TRY(hipStreamSynchronize(st));
    if (h_best) *h_best = tmp[0];
    if (h_kappa) *h_kappa = tmp[1];
    if (h_best_idx) memcpy(h_best_idx, &tmp[8], sizeof(int64_t));
    return 0;
}

template <class M, bool SHIFT, bool SAFE>
static void launch_sweep_spt(int spt, unsigned grid, const SweepArgs& a, hipStream_t st) {
    switch (spt) {
        case 8: sweep_kernel<M, 8, SHIFT, SAFE><<<grid, kBlock, 0, st>>>(a); break;
        case 4: sweep_kernel<M, 4, SHIFT, SAFE><<<grid, kBlock, 0, st>>>(a); break;
        case 2: sweep_kernel<M, 2, SHIFT, SAFE><<<grid, kBlock, 0, st>>>(a); break;
        default: sweep_kernel<M, 1, SHIFT, SAFE><<<grid, kBlock, 0, st>>>(a); break;
    }
}

// every draw packed once per sweep call
template <class M>
static int launch_pack(const SweepArgs& a, hipStream_t st) {
    sweep_pack_kernel<M><<<stream_blocks(a.nd, kBlock), kBlock, 0, st>>>(a);
    OBE_CHECK_LAUNCH("sweep_pack_kernel");
    return 0;
}

template <class M>
static int launch_sweep(const SweepPlan& p, SweepArgs& a, int flags, hipStream_t st) {
    a.tiles_x = p.tiles_x;
    a.nchunks = p.nchunks;
    a.one = 1;
    a.xcd_map = p.nchunks % 8 == 0;
    const unsigned grid = (unsigned)p.tiles_x * (unsigned)p.nchunks;
    const bool shifted = flags & OBE_SWEEP_SHIFTED;
    bool safe = false;
    if constexpr (has_safe_eval<M>::value) safe = flags & OBE_SWEEP_SAFE;
    if (safe) {
        if constexpr (has_safe_eval<M>::value) {        // always shifted: the careful variant
            launch_sweep_spt<M, true, true>(p.spt, grid, a, st);
        }
    } else if (shifted) {
        launch_sweep_spt<M, true, false>(p.spt, grid, a, st);
    } else {
        launch_sweep_spt<M, false, false>(p.spt, grid, a, st);
    }
    OBE_CHECK_LAUNCH("sweep_kernel");
    return 0;
}

static int prepare_sweep(const obe_model* m, obe_model& mm, const double* d_settings, int64_t ld_s, int64_t ns,
                         const double* d_particles, int64_t ld_p, int64_t np, const double* d_weights,
                         const int64_t* d_draw_idx, int64_t n_draws, const double* d_moments, void* d_ws,
                         int64_t ws_bytes, SweepPlan& plan, SweepArgs& a, SweepWs& w, int64_t* ws_need = nullptr) {
    if (!m || !d_settings || !d_particles || !d_moments || ns <= 0 || np <= 0) return bad_arg("sweep: bad pointer/size");
    if (!d_draw_idx && !d_weights) return bad_arg("sweep: full mode needs weights");
    mm = *m;
    if (int rc = obe_model_validate(&mm)) return rc;
    const int64_t nd = d_draw_idx ? n_draws : np;
    if (nd <= 0) return bad_arg("sweep: n_draws must be positive");
    int packed_w = 0, cost = 1;
    if (int rc = dispatch_model(mm, [&](auto M) -> int {
            packed_w = packed_width<decltype(M)>();
            cost = sweep_cost<decltype(M)>::value;
            return 0;
        }))
        return rc;
    plan = plan_sweep(ns, nd, cost, mm.n_channels > 4 ? 2 : 8);
    const int64_t part = (int64_t)plan.nchunks * mm.n_channels * ns;
    if (int rc = carve_sweep_ws(d_ws, ws_bytes, part, (int64_t)mm.n_channels * ns, w, argmax_slots(ns), nd * packed_w))
        return rc;
    if (ws_need) *ws_need = sweep_ws_bytes(part, (int64_t)mm.n_channels * ns, argmax_slots(ns), nd * packed_w);
    a.cs_out = w.cs;
    a.packed = w.packed;
    a.m = mm;
    a.settings = d_settings;
    a.ld_s = ld_s;
    a.ns = ns;
    a.particles = d_particles;
    a.ld_p = ld_p;
    a.weights = d_weights;
    a.draw_idx = d_draw_idx;
    a.nd = nd;
    a.n_particles = np;
    a.uniform_w = 1.0 / (double)nd;
    a.moments = d_moments;
    a.chunk = plan.chunk;
    a.part1 = w.part1;
    a.part2 = w.part2;
    return 0;
}

}  // namespace obe

using namespace obe;

extern "C" {

int obe_sweep_settings_per_lane(int64_t n_settings) {
    return plan_sweep(n_settings < 1 ? 1 : n_settings, (int64_t)1 << 40).spt;      // the most any draw count gets
}

int obe_sweep_settings_per_lane_for(int64_t n_settings, int64_t n_draws) {
    if (n_settings < 1) n_settings = 1;
    if (n_draws < 1) return obe_sweep_settings_per_lane(n_settings);
    static const bool no_small = getenv("OBE_SWEEP_NO_SMALL") != nullptr;
    if (!no_small && n_draws <= kSmallSweepDraws && n_settings * n_draws <= kSmallSweepEvals) return 1;   // one-workgroup path
    return plan_sweep(n_settings, n_draws).spt;
}

int64_t obe_workspace_bytes(int64_t n_particles, int64_t n_settings, int32_t n_channels, int32_t n_dims) {
    if (n_particles < 1) n_particles = 1;
    if (n_settings < 1) n_settings = 1;
    if (n_channels < 1) n_channels = 1;
    if (n_dims < 1) n_dims = 1;
#ifdef OBE_PLUGIN_MODEL_HEADER
    const int packed_w = kMaxPackedWidth;
#else
    const int packed_w = max_packed_width(n_dims);
#endif
    int64_t d = sweep_ws_doubles(n_settings, n_particles, n_channels, packed_w);
    const int64_t nv = std::max<int64_t>(2 + 2 * n_dims, (int64_t)n_dims * (n_dims + 1) / 2);
    d = std::max<int64_t>(d, 1024 * nv + nv);                       // moments (grid cap <= 1024)
    d = std::max<int64_t>(d, 2 * kMaxBlocks + 8 + 1024 * (2 + 2 * (int64_t)n_dims));   // update + fused first moments
    d = std::max<int64_t>(d, 2 * ((n_particles + 2047) / 2048) + 8);      // cdf block sums and minima
    d = std::max<int64_t>(d, 2 * (int64_t)kMaxBlocks + 8);          // update partials
    return (d + 64) * (int64_t)sizeof(double);
}

int obe_sweep_utility(const obe_model* m, const double* d_settings, int64_t ld_s, int64_t n_settings,
                      const double* d_particles, int64_t ld_p, int64_t n_particles, const double* d_weights,
                      const int64_t* d_draw_idx, int64_t n_draws, const double* d_moments, int32_t shifted,
                      const double* d_noise_var, int64_t noise_ld, const double* d_cost, double cost_scalar,
                      double* d_yvar, double* d_utility, double* h_best, int64_t* h_best_idx, double* h_kappa,
                      void* d_ws, int64_t ws_bytes, void* stream) {
    if (!d_noise_var || !d_yvar || !d_utility) return bad_arg("obe_sweep_utility: bad output/noise pointer");
    obe_model mm;
    SweepPlan plan;
    SweepArgs a{};
    SweepWs w;
    int64_t sweep_ws_need = 0;
    if (int rc = prepare_sweep(m, mm, d_settings, ld_s, n_settings, d_particles, ld_p, n_particles, d_weights,
                               d_draw_idx, n_draws, d_moments, d_ws, ws_bytes, plan, a, w, &sweep_ws_need))
        return rc;
    hipStream_t st = as_stream(stream);
    UtilArgs ua;
    if (int rc = make_util_args(ua, d_noise_var, noise_ld, d_cost, cost_scalar, mm.n_channels, mm.n_params)) return rc;
    const bool speculative = shifted & OBE_SWEEP_SPECULATIVE;
    const bool nowait = speculative || (shifted & OBE_SWEEP_NOWAIT);
    if (nowait) {
        if (d_draw_idx) return bad_arg("obe_sweep_utility: OBE_SWEEP_SPECULATIVE / OBE_SWEEP_NOWAIT are for full sweeps");
        if (speculative) {
            if (ws_bytes < sweep_ws_need + 16)
                return bad_arg("obe_sweep_utility: OBE_SWEEP_SPECULATIVE needs 16 spare bytes at the end of the workspace (OBE_WS_ABORT_WORD)");
            a.abort = ws_abort_word(d_ws, ws_bytes);
        }
        if ((h_best && !device_view_of_host(h_best)) || (h_best_idx && !device_view_of_host(h_best_idx)) ||
            (h_kappa && !device_view_of_host(h_kappa)))
            return bad_arg("obe_sweep_utility: OBE_SWEEP_SPECULATIVE / OBE_SWEEP_NOWAIT need page-locked host outputs");
    }
    HostResult hr = host_result(h_best, h_best_idx, h_kappa);
    hr.tail = result_tail(d_ws, ws_bytes, sweep_ws_need);
    static const bool no_small = getenv("OBE_SWEEP_NO_SMALL") != nullptr;      // test / tuning aid
    if (d_draw_idx && !no_small && n_draws <= kSmallSweepDraws && n_settings * n_draws <= kSmallSweepEvals) {
        int rc = dispatch_model(mm, [&](auto M) -> int {
            using Model = decltype(M);
            constexpr int NPKW = (Model::NPK + 2) & ~1;
            const size_t lds = (size_t)n_draws * NPKW * sizeof(double);
            bool safe = false;
            if constexpr (has_safe_eval<Model>::value) safe = shifted & OBE_SWEEP_SAFE;
            if (safe) {
                if constexpr (has_safe_eval<Model>::value)
                    sweep_small_kernel<Model, true><<<1, kBlock, lds, st>>>(a, ua, d_yvar, d_utility, w.out_v, w.out_i, hr);
            } else {
                sweep_small_kernel<Model, false><<<1, kBlock, lds, st>>>(a, ua, d_yvar, d_utility, w.out_v, w.out_i, hr);
            }
            OBE_CHECK_LAUNCH("sweep_small_kernel");
            return 0;
        });
        if (rc) return rc;
        return read_best(w, h_best, h_best_idx, st, h_kappa, hr);
    }
    // (a call without host outputs — a sharded rank reads the record itself — is timed too: it waits for the
    // sweep kernel's end event instead of for the result)
    const bool timed = g_timing.on;
    if (timed) {
        resolve_pending_timing();
        int dev = 0;
        OBE_HIP_TRY(hipGetDevice(&dev));
        if (g_timing.device != dev) {
            if (g_timing.e0) (void)hipEventDestroy(g_timing.e0);
            if (g_timing.e1) (void)hipEventDestroy(g_timing.e1);
            OBE_HIP_TRY(hipEventCreate(&g_timing.e0));
            OBE_HIP_TRY(hipEventCreate(&g_timing.e1));
            g_timing.device = dev;
        }
    }
    int rc = dispatch_model(mm, [&](auto M) -> int {
        if (int e = launch_pack<decltype(M)>(a, st)) return e;
        if (timed) (void)hipEventRecord(g_timing.e0, st);
        const int e = launch_sweep<decltype(M)>(plan, a, shifted, st);
        if (timed) (void)hipEventRecord(g_timing.e1, st);
        return e;
    });
    if (rc) return rc;
    int nb = static_cast<int>((n_settings + kFinSettings - 1) / kFinSettings);
    if (nb > kFinMaxBlocks) return bad_arg("obe_sweep_utility: more than 4 194 304 settings per call");
    static const bool no_narrow = getenv("OBE_FINALIZE_NARROW") && atoi(getenv("OBE_FINALIZE_NARROW")) == 0;   // A/B
    if (plan.nchunks >= kFinManyChunks && nb < kFinNarrowBelow && !no_narrow) {
        nb = static_cast<int>((n_settings + kFinNarrow - 1) / kFinNarrow);       // (< 512: within the argmax slots)
        sweep_finalize<kFinGroupsMany, kFinNarrow><<<nb, kFinGroupsMany * kFinNarrow, 0, st>>>(
            w.part1, w.part2, plan.nchunks, mm.n_channels, n_settings, d_moments, d_draw_idx == nullptr, ua, w.cs,
            d_yvar, d_utility, w.bv, w.bi, w.bk, a.abort);
    } else if (plan.nchunks >= kFinManyChunks)
        sweep_finalize<kFinGroupsMany><<<nb, kFinGroupsMany * kWave, 0, st>>>(
            w.part1, w.part2, plan.nchunks, mm.n_channels, n_settings, d_moments, d_draw_idx == nullptr, ua, w.cs,
            d_yvar, d_utility, w.bv, w.bi, w.bk, a.abort);
    else
        sweep_finalize<kFinGroupsFew><<<nb, kFinGroupsFew * kWave, 0, st>>>(
            w.part1, w.part2, plan.nchunks, mm.n_channels, n_settings, d_moments, d_draw_idx == nullptr, ua, w.cs,
            d_yvar, d_utility, w.bv, w.bi, w.bk, a.abort);
    OBE_CHECK_LAUNCH("sweep_finalize");
    argmax_fold<<<1, kBlock, 0, st>>>(w.bv, w.bi, nb, w.bk, w.out_v, w.out_i, hr, a.abort);
    OBE_CHECK_LAUNCH("argmax_fold");
    if (nowait) {               // nobody waits here: the caller watches the armed words when it wants the result
        if (timed) {
            g_timing.pending = true;
            g_timing.pending_min_ms = 0.5 * (double)n_settings * (double)n_particles / 8e9;
        }
        return 0;
    }
    rc = read_best(w, h_best, h_best_idx, st, h_kappa, hr);
    if (timed && !rc && !(h_best || h_best_idx || h_kappa)) rc = (int)hipEventSynchronize(g_timing.e1);
    if (timed && !rc) {                       // the stream is drained: both events have completed
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_timing.e0, g_timing.e1) == hipSuccess) {
            g_timing.total_ms += ms;
            g_timing.launches += 1;
        }
    }
    return rc;
}

int obe_sweep_timing(int32_t enable, double* h_total_ms, int64_t* h_launches) {
    resolve_pending_timing();
    if (h_total_ms) *h_total_ms = g_timing.total_ms;
    if (h_launches) *h_launches = g_timing.launches;
    if (enable >= 0) {
        g_timing.on = enable != 0;
        g_timing.total_ms = 0.0;
        g_timing.launches = 0;
    }
    return 0;
}

int obe_sweep_kernel_time(const obe_model* m, const double* d_settings, int64_t ld_s, int64_t n_settings,
                          const double* d_particles, int64_t ld_p, int64_t n_particles, const double* d_weights,
                          const double* d_moments, int32_t shifted, void* d_ws, int64_t ws_bytes, int32_t iters,
                          float* h_ms_avg, void* stream) {
    if (!h_ms_avg || iters == 0) return bad_arg("obe_sweep_kernel_time: bad arguments");
    obe_model mm;
    SweepPlan plan;
    SweepArgs a{};
    SweepWs w;
    if (int rc = prepare_sweep(m, mm, d_settings, ld_s, n_settings, d_particles, ld_p, n_particles, d_weights,
                               nullptr, 0, d_moments, d_ws, ws_bytes, plan, a, w))
        return rc;
    hipStream_t st = as_stream(stream);
    hipEvent_t e0, e1;
    OBE_HIP_TRY(hipEventCreate(&e0));
    OBE_HIP_TRY(hipEventCreate(&e1));
    const int sh = shifted;
    int rc = dispatch_model(mm, [&](auto M) -> int { return launch_pack<decltype(M)>(a, st); });
    if (rc) {
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        return rc;
    }
    if (iters < 0) {
        // isolated launches, as a measurement cycle issues them: the stream is drained before each
        // one, so neither the previous launch's tail nor its clocks carry over
        static const int gap_us = getenv("OBE_TIME_GAP_US") ? atoi(getenv("OBE_TIME_GAP_US")) : 0;   // tuning aid
        double total = 0.0;
        for (int i = 0; i < -iters && !rc; ++i) {
            hipError_t e = hipStreamSynchronize(st);
            if (gap_us > 0) {
                timespec ts{0, gap_us * 1000L};
                nanosleep(&ts, nullptr);
            }
            (void)hipEventRecord(e0, st);
            rc = dispatch_model(mm, [&](auto M) -> int { return launch_sweep<decltype(M)>(plan, a, sh, st); });
            (void)hipEventRecord(e1, st);
            if (e == hipSuccess) e = hipEventSynchronize(e1);
            float ms = 0.f;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
            if (e != hipSuccess && !rc) rc = fail(e, "sweep timing");
            total += ms;
        }
        *h_ms_avg = (float)(total / (double)(-iters));
    } else {
        rc = dispatch_model(mm, [&](auto M) -> int { return launch_sweep<decltype(M)>(plan, a, sh, st); });   // warm
        if (!rc) {
            (void)hipEventRecord(e0, st);
            for (int i = 0; i < iters && !rc; ++i)
                rc = dispatch_model(mm, [&](auto M) -> int { return launch_sweep<decltype(M)>(plan, a, sh, st); });
            (void)hipEventRecord(e1, st);
            hipError_t e = hipEventSynchronize(e1);
            float ms = 0.f;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
            if (e != hipSuccess) rc = fail(e, "sweep timing");
            *h_ms_avg = ms / (float)iters;
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

int obe_yspace_variance(const double* d_yspace, int64_t n_draws, int32_t n_channels, int64_t n_settings,
                        double* d_yvar, void* stream) {
    if (!d_yspace || !d_yvar || n_draws <= 0 || n_channels < 1 || n_settings <= 0)
        return bad_arg("obe_yspace_variance: bad pointer/size");
    const int64_t row = (int64_t)n_channels * n_settings;
    yspace_var_kernel<<<stream_blocks(row, kBlock), kBlock, 0, as_stream(stream)>>>(d_yspace, n_draws, row, d_yvar);
    OBE_CHECK_LAUNCH("yspace_var_kernel");
    return 0;
}

int obe_utility_argmax(const double* d_yvar, int32_t n_channels, int64_t n_settings, const double* d_noise_var,
                       int64_t noise_ld, const double* d_cost, double cost_scalar, double* d_utility,
                       double* h_best, int64_t* h_best_idx, void* d_ws, int64_t ws_bytes, void* stream) {
    if (!d_yvar || !d_noise_var || !d_utility || n_settings <= 0 || n_channels < 1)
        return bad_arg("obe_utility_argmax: bad pointer/size");
    SweepWs w;
    if (int rc = carve_sweep_ws(d_ws, ws_bytes, 0, 0, w)) return rc;
    hipStream_t st = as_stream(stream);
    if (noise_ld < 0) return bad_arg("obe_utility_argmax: noise_ld < 0 (noise variance from a K3 block) is for obe_sweep_utility");
    UtilArgs ua;
    if (int rc = make_util_args(ua, d_noise_var, noise_ld, d_cost, cost_scalar, 0, 0)) return rc;
    const int nb = stream_blocks(n_settings, kBlock);
    utility_kernel<<<nb, kBlock, 0, st>>>(d_yvar, n_channels, n_settings, ua, d_utility, w.bv, w.bi);
    OBE_CHECK_LAUNCH("utility_kernel");
    HostResult hr = host_result(h_best, h_best_idx, nullptr);
    hr.tail = result_tail(d_ws, ws_bytes, sweep_ws_bytes(0, 0, kMaxBlocks, 0));
    argmax_fold<<<1, kBlock, 0, st>>>(w.bv, w.bi, nb, nullptr, w.out_v, w.out_i, hr);
    OBE_CHECK_LAUNCH("argmax_fold");
    return read_best(w, h_best, h_best_idx, st, nullptr, hr);
}

int obe_argmax(const double* d_v, int64_t n, double* h_best, int64_t* h_best_idx, void* d_ws, int64_t ws_bytes,
               void* stream) {
    if (!d_v || n <= 0) return bad_arg("obe_argmax: bad pointer/size");
    SweepWs w;
    if (int rc = carve_sweep_ws(d_ws, ws_bytes, 0, 0, w)) return rc;
    hipStream_t st = as_stream(stream);
    const int nb = stream_blocks(n, kBlock);
    argmax_kernel<<<nb, kBlock, 0, st>>>(d_v, n, w.bv, w.bi);
    OBE_CHECK_LAUNCH("argmax_kernel");
    HostResult hr = host_result(h_best, h_best_idx, nullptr);
    hr.tail = result_tail(d_ws, ws_bytes, sweep_ws_bytes(0, 0, kMaxBlocks, 0));
    argmax_fold<<<1, kBlock, 0, st>>>(w.bv, w.bi, nb, nullptr, w.out_v, w.out_i, hr);
    OBE_CHECK_LAUNCH("argmax_fold");
    return read_best(w, h_best, h_best_idx, st, nullptr, hr);
}

}  // extern "C"
