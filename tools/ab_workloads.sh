for wl in evo_eval_sr_small sr_small_supernet; do
  export BENCH_FLAGS="--workload $wl"
  echo "== $wl"
  bash tools/ab.sh "VITRES_NT_WIDE=0" "VITRES_NT_WIDE=1 VITRES_NTW_SK=0" "VITRES_NT_WIDE=1 VITRES_NTW_SK=1"
done
