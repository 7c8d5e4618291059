/* tests/test_device_choice_cpu.py: the device logic of rnnoise_batch_create / the pools behind rnnoise_create (rnnoise_amd/csrc/
 * device_choice.h) against stubbed device counts.  argv: visible pinned n_pools  ->  prints the device of each pool, then the
 * verdict on device indices -1 .. visible. */
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../rnnoise_amd/csrc/device_choice.h"
int main(int argc, char **argv) {
  if (argc != 4) return 2;
  const int visible = atoi(argv[1]), pinned = atoi(argv[2]), n = atoi(argv[3]);
  for (int k = 0; k < n; k++) printf("%d ", rn_pool_device((size_t)k, pinned, visible));
  printf("|");
  for (int d = -1; d <= visible; d++) printf(" %d", rn_device_index_ok(d, visible) ? 1 : 0);
  printf("\n");
  return 0;
}
