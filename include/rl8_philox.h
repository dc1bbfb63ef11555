/*
 * rl8_philox.h -- the noise specification of the MI355X PPO path.
 *
 * The reference draws all randomness from torch's default generator
 * (DummyEnv.reset `uniform_` src/rl8/env.py:200-202, Categorical/Normal `sample`
 * src/rl8/distributions.py:121-122, Batcher `randperm` src/rl8/_utils.py:214).
 * A CPU mt19937 stream cannot be reproduced on a GPU, so this build defines its
 * own counter-based noise: Philox4x32-10 (Salmon, Moraes, Dror, Shaw, "Parallel
 * Random Numbers: As Easy as 1, 2, 3", SC'11), keyed by a 64-bit seed and
 * addressed by (row, step, stream, block).  The same header is compiled into the
 * HIP kernels (device) and into the CPU oracle (host), so both produce the same
 * words; transcendental transforms of those words are evaluated in fp64 and
 * rounded once to fp32 so host and device agree bit-for-bit in practice.
 *
 * Parity with the reference itself is checked by INJECTING the noise the
 * reference drew (recorded in tests/golden/), not by matching generators.
 */
#ifndef RL8_PHILOX_H
#define RL8_PHILOX_H

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define RL8_HD __host__ __device__ __forceinline__
#else
#define RL8_HD static inline
#endif

/* Streams keep draws for different purposes independent. */
#define RL8_STREAM_RESET 1u
#define RL8_STREAM_ACTION 2u
#define RL8_STREAM_SHUFFLE 3u

RL8_HD void rl8_philox4x32_10(uint64_t seed, uint64_t row, uint64_t step, uint32_t stream_block,
                              uint32_t out[4]) {
  uint32_t c0 = (uint32_t)row, c1 = (uint32_t)(row >> 32);
  uint32_t c2 = (uint32_t)step, c3 = stream_block;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* stream in the top byte, 24-bit block below it. */
RL8_HD uint32_t rl8_stream_block(uint32_t stream, uint32_t block) {
  return (stream << 24) | (block & 0x00FFFFFFu);
}

/* 24-bit uniform in [0, 1): exact in fp32. */
RL8_HD float rl8_u01_24(uint32_t r) { return (float)(r >> 8) * 5.9604644775390625e-08f; }

/* 24-bit uniform in (0, 1) as a double: (k + 0.5) * 2^-24. */
RL8_HD double rl8_u01_open(uint32_t r) {
  return ((double)(r >> 8) + 0.5) * 5.9604644775390625e-08;
}

/* Exp(1) draw number `w` of (row, step): q = -log(u), fp64 then one rounding. */
RL8_HD float rl8_exponential(uint64_t seed, uint64_t row, uint64_t step, uint32_t w) {
  uint32_t r[4];
  rl8_philox4x32_10(seed, row, step, rl8_stream_block(RL8_STREAM_ACTION, w >> 2), r);
  return (float)(-log(rl8_u01_open(r[w & 3u])));
}

/* Box-Muller on two words, fp64 then one rounding each. */
RL8_HD void rl8_box_muller(uint32_t r0, uint32_t r1, float *z0, float *z1) {
  const double u1 = rl8_u01_open(r0), u2 = rl8_u01_open(r1);
  const double radius = sqrt(-2.0 * log(u1));
  const double angle = 6.283185307179586476925286766559 * u2;
  *z0 = (float)(radius * cos(angle));
  *z1 = (float)(radius * sin(angle));
}

/* N(0,1) draw number `w` of (row, step). */
RL8_HD float rl8_normal(uint64_t seed, uint64_t row, uint64_t step, uint32_t w) {
  uint32_t r[4];
  float z0, z1;
  const uint32_t pair = w >> 1;
  rl8_philox4x32_10(seed, row, step, rl8_stream_block(RL8_STREAM_ACTION, pair >> 1), r);
  rl8_box_muller(r[(pair & 1u) * 2u], r[(pair & 1u) * 2u + 1u], &z0, &z1);
  return (w & 1u) ? z1 : z0;
}

#endif /* RL8_PHILOX_H */
