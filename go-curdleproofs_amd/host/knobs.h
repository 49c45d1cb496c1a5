// The library's tunables ("knobs") in ONE table.  Every knob is read ONCE from the environment
// (CURDLE_<NAME>), the first time any knob is asked for -- in practice at curdle_init -- and never
// again: getenv beside a host program's setenv is not thread-safe, and a dozen of them on every MSM
// was what INTEGRATION.md said did not happen.  After that a value changes only through
// curdle_plan_override(name, value), the hook the tests and the measurement tools use to walk a
// knob through its range inside one process.
//
// Knobs whose experiments are closed were removed in round 4 (the measurements stay under
// profiles/): CURDLE_TWO_ROUNDS, CURDLE_TAIL_PRIO, CURDLE_SYNC_STREAMS, CURDLE_PRE_STREAMS,
// CURDLE_SEG_LEN_MIN, CURDLE_SORT_CHUNK, CURDLE_FUSE_SCAN_MAX, CURDLE_HOST_ONE_COPY (=
// CURDLE_HOST_CHUNKS=1), CURDLE_DECODE_PRIO, CURDLE_BATCH_EXTRA_PRODUCERS, CURDLE_NO_IFMA.
// Round 6, the same for the experiments rounds 4 and 5 closed: CURDLE_ROUND_LANES, CURDLE_SYNC_LANES, CURDLE_PIPE_LANES,
// CURDLE_REDUCE_BITS, CURDLE_GPU_COMBINE_MIN, CURDLE_SCAN (with k_scan_fused), CURDLE_FRONT, CURDLE_DIRECT_RESULTS,
// CURDLE_HOST_OVERLAP_MIN, CURDLE_HOST_SORT_STREAMS, CURDLE_HOST_GRADED, CURDLE_HOST_PATTERN.
#pragma once
#include <stddef.h>

namespace curdle {
namespace knobs {
enum Id {
  WINDOW_BITS,        // maximum window width c of every plan (4..16); unset: the size table
  SEG_LEN,            // sorted positions per accumulate lane
  REDUCE_SEG,         // buckets per bucket-reduce segment (power of two)
  SCATTER,            // 1: one-pass scatter, 2: two-pass wherever the shapes allow; unset: by size
  HOST_CHUNKS,        // chunks of a host-buffer MSM from 2^19 pairs (1..4)
  MAX_MSMS_PER_PASS,  // MSMs per pass of a batch beyond the bucket-slot limit
  MULTI_DEVICE_MIN,   // pairs from which curdle_msm_g1 spreads over the configured devices
  MAIN_STREAMS,       // accumulate streams of a context (1..4), read when the context is created
  TWO_KERNEL_MAX,     // one-shot decodings up to this many points take the two-kernel form
  QUAD_MAX_LANES,     // decode / scalar-multiplication kernels: quads up to this many lanes
  BATCH_CHUNK,        // batch verification: proofs per decode-ahead chunk
  BATCH_PRODUCERS,    // ... decode-ahead producer threads
  BATCH_GROUP,        // ... proofs settled per device accumulation
  DEVICE_ACC,         // 0: the verifier's accumulator on the host mirror
  HOST_DECODE,        // 1: point decoding on the host
  VERIFY_EAGER,       // 1: check points evaluated where the reference does
  VERIFY_TRACE,       // 1: per-phase timings of a verification on stderr
  PROVER_FOLD_BASES,  // 1: the prover folds its bases round by round like the reference
  ACC_PRIO,           // k_accumulate: the two waves of a SIMD take turns at high priority every 2^v x 10 ns (0: never; unset: 15 for synchronous calls from half a round of lanes)
  REDUCE_PRIO,        // wave priority (0..3) of k_reduce_segments / k_reduce_level; unset: 3 for pipelined calls, 0 for synchronous ones
  AUX_PRIO,           // wave priority (0..3) of the sort kernels, the conversion and the chunk fold; unset: 3 for pipelined calls, 0 for synchronous ones
  HOST_FOLD,          // 0: chunked host-buffer MSMs keep every chunk's fragments for the one reduction (no progressive folding)
  COUNT
};
// The knob's value, or -1 if it is not set (every knob's valid values are >= 0).
long long get(Id id);
inline bool is_set(Id id) { return get(id) >= 0; }
// name: "WINDOW_BITS" or "CURDLE_WINDOW_BITS"; value < 0 unsets.  0 on success, -1 for an unknown name.
int set(const char* name, long long value);
const char* name(Id id);
}  // namespace knobs
}  // namespace curdle
