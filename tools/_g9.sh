export SMG_EXPECT_GPU=1
python tools/fuzz_parity.py 60 3000 2>&1 | tail -25
python tools/fuzz_reprecompute.py 2>&1 | tail -4
