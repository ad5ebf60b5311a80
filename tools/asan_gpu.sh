#!/bin/bash
# Sanitizer lane on a GPU box: the host C++ of libsmg (ASan + UBSan build, build.py: build_sanitized) under the GPU parity tests, so that
# the host code that only runs with a device (SELL uploads, recipes, solve orchestration, graph capture) is covered too.
#   tools/asan_gpu.sh [pytest args]
python - <<'PY'
from surface_multigrid_code_amd import build as b
print(b.build_sanitized())
PY
eval "$(python - <<'PY'
from surface_multigrid_code_amd import build as b
e = b.sanitizer_env({})
for k, v in e.items():
    print("export %s='%s'" % (k, v))
PY
)"
export SMG_EXPECT_GPU=1
# (tests that initialise torch.cuda cannot run with libasan preloaded -- torch's lazy dlopen of its NVRTC shim fails -- and are deselected)
timeout 1500 python -m pytest ${@:-tests/test_gpu_parity.py tests/test_gpu_smoothers.py} -q -p no:cacheprovider \
  -k "not split_phase and not speculative and not allreduce and not mean_curvature_flow_steps and not device_assembly and not solve_sharded" > gpurun_out/asan_gpu.log 2>&1
grep -c "AddressSanitizer\|runtime error:" gpurun_out/asan_gpu.log | sed 's/^/sanitizer reports: /'
grep "AddressSanitizer\|runtime error:" gpurun_out/asan_gpu.log | head -5
tail -3 gpurun_out/asan_gpu.log
