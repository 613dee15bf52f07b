// tools/probes/edge2_phases.hpp — cycle counters between the phases of a step of k_edge2, written behind the status words for
// tools/gpu_sweep.py (BT_DEBUG_MODE=64 prints them); every probe waits for the LDS and fences the scheduler.
// Build:  tools/build_variant.sh <name> ba_edge2.hip -DBT_PROBE_HEADER='"../../tools/probes/edge2_phases.hpp"'
#pragma once

#define BT_PROBE_E2_DECL() long long pf[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pf_c = clock64(), pf_n
#define BT_PROBE_E2_TILES_DONE() do { } while (0)
#define BT_E2_PF(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); pf_n = clock64(); pf[i] += pf_n - pf_c; pf_c = pf_n; __builtin_amdgcn_sched_barrier(0); } while (0)
#define BT_PROBE_E2_END(gw, nwaves)                                                                                         \
    do {                                                                                                                    \
        BT_E2_PF(8);                                                                                                        \
        if (lane == 0 && ((gw) == 0 || (gw) == (nwaves) / 2)) {                                                             \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                \
            BT_E2_PF(9);                                                                                                    \
            long long *o = reinterpret_cast<long long *>(a.status + 4) + ((gw) == 0 ? 20 : 40);                             \
            for (int i_ = 0; i_ < 10; ++i_) o[i_] = pf[i_];                                                                 \
            o[10] = t_end - t_begin;                                                                                        \
            o[11] = pf[10]; o[12] = pf[11]; o[13] = pf[12];                                                                 \
        }                                                                                                                   \
    } while (0)
