/* CPU oracle (C) for the eeyore MCMC hot path -- TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the library built
 * from this file; nothing under eeyore_amd/ links or calls it.  Parity status: PINNED against the
 * golden vectors captured from the reference (tests/golden/make_golden.py, tests/test_oracle_c.py).
 *
 * Build: see oracle/Makefile  (gcc -O2 -fopenmp -shared -fPIC; no -ffast-math so NaN => reject holds).
 */
#include <math.h>
#include <stdlib.h>
#include <stddef.h>

#define OC_MAX_LAYERS 8

typedef struct {
  int nl;                       /* number of layers K                     */
  int dims[OC_MAX_LAYERS + 1];  /* d_0 .. d_K                             */
  int bias[OC_MAX_LAYERS];      /* per-layer bias flag (mlp.py:40-42)     */
  int acts[OC_MAX_LAYERS];      /* 0 none, 1 sigmoid, 2 tanh, 3 relu      */
  int lik;                      /* 0 BCE-sum on probabilities, 1 CE-sum   */
} oc_spec;

/* Model.num_params, eeyore/models/model.py:34-36 */
int oc_num_params(const oc_spec* s) {
  int P = 0;
  for (int l = 0; l < s->nl; ++l) P += (s->dims[l] + (s->bias[l] ? 1 : 0)) * s->dims[l + 1];
  return P;
}

int oc_work_size(const oc_spec* s) {
  int hsz = 0, dmax = 0;
  for (int l = 0; l <= s->nl; ++l) { hsz += s->dims[l]; if (s->dims[l] > dmax) dmax = s->dims[l]; }
  return hsz + 2 * dmax;
}

#define REAL double
#define SUFFIX f64
#define EXP exp
#define LOG log
#define TANH tanh
#include "mlp_oracle_impl.h"
#undef REAL
#undef SUFFIX
#undef EXP
#undef LOG
#undef TANH

#define REAL float
#define SUFFIX f32
#define EXP expf
#define LOG logf
#define TANH tanhf
#include "mlp_oracle_impl.h"
