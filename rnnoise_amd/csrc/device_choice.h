// device_choice.h -- which GPU a thing goes to, as pure functions of the visible-device count (no HIP in here, so that
// tests/test_device_choice_cpu.py can run them against any count: no round has had more than one GPU to run them on).
#pragma once
#include <stddef.h>

// rnnoise_batch_create(model, n, device): an index the runtime shows, or nothing -- there is no CPU fallback and no silent
// remapping of an out-of-range index onto device 0
static inline bool rn_device_index_ok(int device, int visible) { return device >= 0 && device < visible; }

// The device of the k-th pool (k = pools the model already has) behind rnnoise_create().  The reference lets a process hold any
// number of independent states (include/rnnoise.h:80, src/denoise.c:311-321); on a multi-GPU node a few thousand of them should
// not all land on device 0, so pools go to the visible devices in turn.  pinned >= 0 ($RNNOISE_AMD_DEVICE): that device for every
// pool, clamped to the last visible one (a job script written for an 8-GPU node still runs on a 4-GPU one).  visible < 1 is
// treated as 1: the caller finds out that there is no device when the pool's batch is created.
static inline int rn_pool_device(size_t pools_so_far, int pinned, int visible) {
  const int count = visible < 1 ? 1 : visible;
  if (pinned >= 0) return pinned < count ? pinned : count - 1;
  return (int)(pools_so_far % (size_t)count);
}
