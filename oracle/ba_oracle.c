/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 * CPU oracle for the BA hot path: the restatement in ba_oracle_impl.h compiled
 * for double (…_f64) and float (…_f32).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.
 * Build: make -C oracle   ->  oracle/_build/libba_oracle.so                */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define REAL double
#define SUF _f64
#include "ba_oracle_impl.h"
#undef REAL
#undef SUF

#define REAL float
#define SUF _f32
#include "ba_oracle_impl.h"
#undef REAL
#undef SUF
