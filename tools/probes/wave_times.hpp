// tools/probes/wave_times.hpp — every wave's clock (100 MHz wall_clock64) at the marks of k_edge2 / k_etile / k_tile, written into
// patches_out for tools/gpu_wave_times*.py.  Build:  tools/build_variant.sh <name> <source.hip> -DBT_PROBE_HEADER='"../../tools/probes/wave_times.hpp"'
// (the hooks are named in batrack_amd/csrc/probe.hpp; `a`, `lane` are the kernels' own names).
#pragma once

// ---- k_edge2: start, end of the wave's tiles, end, and where it ran (HW_ID, XCC_ID)
#define BT_PROBE_E2_DECL() const long long wt0 = wall_clock64(); long long wt1 = 0
#define BT_PROBE_E2_TILES_DONE() wt1 = wall_clock64()
#define BT_PROBE_E2_END(gw, nwaves)                                                                                         \
    do {                                                                                                                    \
        if (lane == 0) {                                                                                                    \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                \
            long long *o = reinterpret_cast<long long *>(a.patches_out) + 4 * (size_t)(gw);                                 \
            o[0] = wt0; o[1] = wt1; o[2] = wall_clock64();                                                                  \
            o[3] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) /* HW_ID */ |                                       \
                   ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) /* XCC_ID */ << 32);                              \
        }                                                                                                                   \
    } while (0)

// ---- k_etile: start, after the prologue, the rounds, the merge, the end
#define BT_PROBE_ET_DECL() long long wt[5] = {(long long)wall_clock64(), 0, 0, 0, 0}
#define BT_PROBE_ET_MARK(i) wt[i] = (long long)wall_clock64()
#define BT_PROBE_ET_END(full, lane_, wave_, extra)                                                                              \
    do {                                                                                                                    \
        if ((full) && (lane_) == 0) {                                                                                       \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                \
            wt[4] = (long long)wall_clock64();                                                                              \
            long long *o = reinterpret_cast<long long *>(a.patches_out) + 8 * ((size_t)blockIdx.x * 8 + (wave_));           \
            for (int i_ = 0; i_ < 5; ++i_) o[i_] = wt[i_];                                                                  \
            o[5] = (extra);                                                                                                 \
        }                                                                                                                   \
    } while (0)

// ---- k_tile: start, after the prologue, the slots, the merge + Q, the Schur product, the end
#define BT_PROBE_TILE_DECL() long long wt[6] = {(long long)wall_clock64(), 0, 0, 0, 0, 0}
#define BT_PROBE_TILE_MARK(i) wt[i] = (long long)wall_clock64()
#define BT_PROBE_TILE_END(on, lane_, wave_, nwaves)                                                                         \
    do {                                                                                                                    \
        if ((on) && (lane_) == 0) {                                                                                         \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                \
            wt[5] = (long long)wall_clock64();                                                                              \
            long long *o = reinterpret_cast<long long *>(a.patches_out) + 8 * ((size_t)blockIdx.x * (nwaves) + (wave_));    \
            for (int i_ = 0; i_ < 6; ++i_) o[i_] = wt[i_];                                                                  \
        }                                                                                                                   \
    } while (0)
