export BENCH_FLAGS="--workload evo_eval_sr_small --steps 64 --warmup 8"
bash tools/ab.sh "VITRES_FUSE_LN_MAXN=256" "VITRES_FUSE_LN_MAXN=320" "VITRES_FUSE_LN_MAXN=640"
