#!/bin/bash
# First line of every script that spends GPU budget: one trivial torch op on cuda:0.  Twice in round 4 a box of the pool answered EVERY
# process with "Memory access fault by GPU node-2" (the first tensor.cuda() of the first test); the second time the profile script ran
# on through its per-step timeouts and spent the round's remaining 32 GPU-minutes.   usage: scripts/gpu_probe.sh || exit 3
timeout 180 python - <<'PY'
import torch
x = torch.arange(1024, device="cuda", dtype=torch.float32)
assert float((x * 2).sum()) == 1023 * 1024
print("GPU probe ok:", torch.cuda.get_device_name(0))
PY
rc=$?
[ $rc -ne 0 ] && echo "GPU probe FAILED (rc $rc): this box is not used"
exit $rc
