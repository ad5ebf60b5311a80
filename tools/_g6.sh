export SMG_EXPECT_GPU=1
for p in 32 64 128 256; do echo "== SMG_PITCH_SPEC_MAX=$p"; SMG_PITCH_SPEC_MAX=$p SMG_TOOL_SMOOTHER=hybrid_chebyshev:300000 python tools/level_times.py C3 2>&1 | tail -5; done
