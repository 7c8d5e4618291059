#!/usr/bin/env python3
"""VGPRs / spills / instruction counts of the kernels in a built object (no GPU): tools/kmeta.py nn_layers [name filter]"""
import os
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_kernel_budgets_cpu as t  # noqa: E402

obj = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
build = os.environ.get("KMETA_BUILD", t.BUILD)
meta, code = t._kernels(os.path.join(build, obj + ".o"))
for k, v in meta.items():
    if flt not in k:
        continue
    c = Counter(code.get(k, []))
    n = lambda p: sum(x for i, x in c.items() if i.startswith(p))
    print(f"{k:34s} vgpr {v['vgpr_count']:3d} spill {v['vgpr_spill_count']:3d} scratchB {v['private_segment_fixed_size']:4d} insts {sum(c.values()):5d} "
          f"valu {n('v_') - n('v_mfma'):5d} mfma {n('v_mfma'):4d} ds_read {n('ds_read'):4d} ds_write {n('ds_write'):3d} vmem {n('global_') + n('buffer_'):4d} "
          f"scratch {n('scratch_'):3d} waitcnt {c['s_waitcnt']:4d} barrier {c['s_barrier']:2d}")
