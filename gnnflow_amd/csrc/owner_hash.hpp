// owner(v) = splitmix64(v) mod P — who holds a vertex's edges / a row of a sharded feature
// table (the reference's HashPartitioner uses Python's salted hash(str(v)) % P,
// gnnflow/distributed/partition.py:324, which is not reproducible across processes).
#pragma once

#include <cstdint>

#include "common.hpp"

namespace gf {

// z mod P without the 64-bit division routine (~150 instructions on this ISA): with
// m = floor(2^64 / P), q = mulhi(z, m) is floor(z / P) or one less, so one conditional
// subtraction makes the remainder exact.  P == 1 (m does not fit): owner 0.
struct OwnerDiv { uint32_t P; uint64_t m; };
inline OwnerDiv owner_div(uint32_t P) {
  return OwnerDiv{P, P > 1 ? static_cast<uint64_t>((static_cast<unsigned __int128>(1) << 64) / P) : 0};
}
__device__ inline uint32_t owner_of(int64_t v, OwnerDiv d) {
  uint64_t z = static_cast<uint64_t>(v) + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  if (d.P == 1) return 0;
  uint64_t r = z - __umul64hi(z, d.m) * d.P;
  if (r >= d.P) r -= d.P;
  return static_cast<uint32_t>(r);
}

}  // namespace gf
