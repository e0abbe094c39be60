// knobs.h -- what the library reads from the environment, in one place.
//
// SUPPORTED (documented in INTEGRATION.md section 5; read where they act, with std::getenv):
//   NGSLD_REPLAY, NGSLD_REPLAY_DEVICE, NGSLD_REPLAY_SKIP, NGSLD_REPLAY_THREADS, NGSLD_EXACT_STORE   the exact-order replay
//   NGSLD_PAIR_KERNEL                                                                              kernel family override
//   NGSLD_GZ_LEVEL, NGSLD_HOST_TEXT, NGSLD_PIPELINE, NGSLD_PIN_REGISTER                             the drop-in binary / writer
//   NGSLD_TRACE, NGSLD_TIMING, NGSLD_ROCTX, NGSLD_MULTI_VERBOSE                                     diagnostics
//
// TEST-ONLY fault injectors and shapes (tests/ set them; nothing else should): NGSLD_TEST_<NAME>, read through test_knob("<NAME>").
// They force the paths a healthy box never takes -- a host that cannot pin, a device without room, launches cut into many
// grids, batches of a few hundred rows -- so that the suite can hold those paths to the same records.
//   BATCH_PAIRS, STAGE_BYTES, MAX_BLOCKS, PIN_LIMIT_BYTES, PREP_EXACT, HARD_KERNEL, SLAB_SITES, MULTI_DIST,
//   RUN_DIRECT, RUN_TAPER, RUN_STREAMS, TAIL_LEN, TAIL_PAIRS, TILES, TILE_MIN_MB,
//   TEXT_STREAMS, TEXT_HOST_PATCH, TEXT_HOST_PATCH_FAIL_EVERY, TEXT_FALLBACK_EVERY,
//   TEXT_GROUPS, TEXT_GROUP_PAIRS,
//   EXACT_CHUNK_SITES, EXACT_SLOW_US, EXACT_STORE_NO_ROOM, REPLAY_LIST_CAP, REPLAY_SOURCE, LANE_ITER_CAP
// (The knobs of closed A/B experiments -- lane caps and waves, sort-key tilings, run lengths, tile rows, text batch sizes -- are
// gone; their measurements are in HISTORY.md.)
#pragma once

#include <cstdlib>
#include <cstring>
#include <string>

namespace ngsld {

inline const char *test_knob(const char *name) {
  std::string key = "NGSLD_TEST_";
  key += name;
  return std::getenv(key.c_str());
}
inline bool test_knob_is(const char *name, const char *value) {
  const char *v = test_knob(name);
  return v != nullptr && std::strcmp(v, value) == 0;
}

}  // namespace ngsld
