// probe.hpp — measurement hooks of the kernels.  The product build defines every hook EMPTY; a measurement build
// (tools/build_variant.sh ... -DBT_PROBE_HEADER='"../../tools/probes/<file>.hpp"') includes a header from tools/probes/ that
// gives some of them a body (wave clocks, phase cycle counters) — the measurement code itself lives there, not in the kernels.
#pragma once
#ifdef BT_PROBE_HEADER
#include BT_PROBE_HEADER
#endif
// k_edge2 (ba_edge2.hip)
#ifndef BT_PROBE_E2_DECL
#define BT_PROBE_E2_DECL()                  /* at the kernel's top */
#define BT_PROBE_E2_TILES_DONE()            /* behind the wave's last tile */
#define BT_PROBE_E2_END(gw, nwaves)         /* at the kernel's end */
#endif
#ifndef BT_E2_PF
#define BT_E2_PF(i) do { } while (0)        /* phase boundary i of a step */
#endif
// k_etile (ba_etile.hip)
#ifndef BT_PROBE_ET_DECL
#define BT_PROBE_ET_DECL()
#define BT_PROBE_ET_MARK(i)
#define BT_PROBE_ET_END(full, lane, wave, extra)
#endif
// k_tile (ba_kernels.hip)
#ifndef BT_PROBE_TILE_DECL
#define BT_PROBE_TILE_DECL()
#define BT_PROBE_TILE_MARK(i)
#define BT_PROBE_TILE_END(on, lane, wave, nwaves)
#endif
