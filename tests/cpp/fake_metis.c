/* TEST DOUBLE — not METIS and not part of the product.  tests/test_partitioners.py compiles this into a shared object and points
 * LPMP_METIS_LIB at it to exercise lp_mp_amd/multi_gpu.py's ctypes binding of a libmetis (probe in a child process for the idx_t
 * width, option array, argument order of METIS_PartGraphKway as metis.h 5.x declares it) on a box that has no METIS installed.
 * The "partition" is contiguous chunks of the vertex range; the returned objective is the number of cut edges.
 * Built for either idx_t width: -DIDX_BITS=32 (default) or 64, and for either option layout of metis.h: -DMETIS_LAYOUT=51 (default:
 * NCUTS = 7, SEED = 8, UFACTOR = 16, NUMBERING = 17), 52 (NIPARTS and ONDISK inserted: NCUTS = 8, SEED = 9, UFACTOR = 17,
 * NUMBERING = 18) or 0 (options are not looked at).  Like METIS' own CheckParams it refuses NCUTS <= 0 and a NUMBERING other than
 * 0 / 1, which is how a caller that writes a 5.1 slot into a 5.2 library (or the other way round) finds out.
 * -DFAIL_ABOVE=n: every call on more than n vertices fails (a METIS that is installed and broken). */
#include <stdint.h>
#ifndef IDX_BITS
#define IDX_BITS 32
#endif
#if IDX_BITS == 64
typedef int64_t idx_t;
#else
typedef int32_t idx_t;
#endif
typedef float real_t;
#ifndef METIS_LAYOUT
#define METIS_LAYOUT 51
#endif
#if METIS_LAYOUT == 52
enum { OPT_NCUTS = 8, OPT_SEED = 9, OPT_UFACTOR = 17, OPT_NUMBERING = 18 };
#else
enum { OPT_NCUTS = 7, OPT_SEED = 8, OPT_UFACTOR = 16, OPT_NUMBERING = 17 };
#endif

int METIS_SetDefaultOptions(idx_t* options) {
  for (int i = 0; i < 40; ++i) options[i] = -1;
  return 1; /* METIS_OK */
}

int METIS_PartGraphKway(idx_t* nvtxs, idx_t* ncon, idx_t* xadj, idx_t* adjncy, idx_t* vwgt, idx_t* vsize, idx_t* adjwgt, idx_t* nparts,
                        real_t* tpwgts, real_t* ubvec, idx_t* options, idx_t* objval, idx_t* part) {
  (void)ncon; (void)vwgt; (void)vsize; (void)adjwgt; (void)tpwgts; (void)ubvec;
  const idx_t n = *nvtxs, k = *nparts;
  if (n <= 0 || k <= 0) return -2; /* METIS_ERROR_INPUT */
#ifdef FAIL_ABOVE
  if (n > FAIL_ABOVE) return -4; /* METIS_ERROR */
#endif
#if METIS_LAYOUT != 0
  if (options) {
    if (options[OPT_NCUTS] != -1 && options[OPT_NCUTS] <= 0) return -2;
    if (options[OPT_NUMBERING] != -1 && options[OPT_NUMBERING] != 0 && options[OPT_NUMBERING] != 1) return -2;
    if (options[OPT_UFACTOR] != -1 && options[OPT_UFACTOR] <= 0) return -2;
  }
#else
  (void)options;
#endif
  for (idx_t v = 0; v < n; ++v) part[v] = (idx_t)(((int64_t)v * k) / n);
  idx_t cut = 0;
  for (idx_t v = 0; v < n; ++v)
    for (idx_t e = xadj[v]; e < xadj[v + 1]; ++e)
      if (adjncy[e] < 0 || adjncy[e] >= n) return -2;
      else if (adjncy[e] > v && part[adjncy[e]] != part[v]) ++cut;
  *objval = cut;
  return 1;
}
