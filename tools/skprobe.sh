python tools/gemm_wide_bench.py s2 s3 2>&1 | grep -v amdgpu.ids
echo "== no SK"; VITRES_NTW_SK=0 python tools/gemm_wide_bench.py s2 s3 2>&1 | grep -v amdgpu.ids
for sk in 1 0; do for c in "8320 512 1536 res" "8192 2048 512 fwd"; do VITRES_NTW_SK=$sk python tools/ntw_stamps.py $c 2>&1 | grep -v amdgpu.ids | head -6; done; done
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "gemm" 2>&1 | tail -3
