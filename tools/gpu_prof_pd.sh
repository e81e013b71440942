export SEHIP_LIB=$PWD/semantic-embeddings_amd/sehip/libsehip_tuning.so
SE_PD_PROFILE=1 timeout 300 python tools/bench_kernels.py pdist --reps 1 2>&1 | grep -v amdgpu.ids | grep "profile\|^pdist" | head -12
SE_PD_PROFILE=1 timeout 300 python tools/bench_kernels.py shard --reps 2 --q 8192 2>&1 | grep "profile\|^shard" | head -8
