/*
 * TEST INFRASTRUCTURE ONLY - CPU oracle (plain C) for the kornia warp + filter hot path.
 * See ko_impl.h for the per-function reference citations.  Built by oracle/Makefile into
 * oracle/_build/libkornia_oracle.so and loaded by oracle/oracle.py (ctypes + numpy).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define REAL float
#define SFX f32
#define KO_IS_DOUBLE 0
#include "ko_impl.h"
#undef REAL
#undef SFX
#undef KO_IS_DOUBLE

#define REAL double
#define SFX f64
#define KO_IS_DOUBLE 1
#include "ko_impl.h"
#undef REAL
#undef SFX
#undef KO_IS_DOUBLE

int ko_abi_version(void) { return 1; }
