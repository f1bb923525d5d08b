/*
 * gnnflow_rng.h — counter-based RNG shared by the HIP uniform sampler and the
 * CPU oracle.
 *
 * The reference draws `curand(&state[tid]) % num_candidates` from per-thread
 * XORWOW states (gnnflow/csrc/sampling_kernels.cu:202, states initialised by
 * gnnflow/csrc/utils.cu:88-94 with curand_init(seed, tid, 0)).  cuRAND's XORWOW
 * skip-ahead tables are not available off CUDA, so uniform sampling is defined
 * here as "distribution-matched", not bit-matched, to the reference: slot `tid`
 * of the `call`-th sample_layer invocation of a sampler draws
 *
 *     philox4x32_10(counter = {tid_lo, tid_hi, call_lo, call_hi},
 *                   key     = {seed_lo, seed_hi})[0] % num_candidates
 *
 * which keeps the reference's properties (independent stream per (root, slot),
 * sampling with replacement, advancing on every call, same modulo bias) and is
 * stateless, so no RNG state array lives in HBM.
 *
 * Philox4x32-10: Salmon et al., "Parallel Random Numbers: As Easy as 1, 2, 3",
 * SC'11 (public algorithm; constants from the paper).
 */
#ifndef GNNFLOW_RNG_H_
#define GNNFLOW_RNG_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define GF_RNG_FN __host__ __device__ static inline
#else
#define GF_RNG_FN static inline
#endif

GF_RNG_FN uint32_t gf_philox4x32_10_first(uint64_t seed, uint64_t tid,
                                          uint64_t call) {
  uint32_t c0 = (uint32_t)tid, c1 = (uint32_t)(tid >> 32);
  uint32_t c2 = (uint32_t)call, c3 = (uint32_t)(call >> 32);
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c0;
}

#endif  /* GNNFLOW_RNG_H_ */
