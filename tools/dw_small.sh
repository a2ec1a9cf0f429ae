# developer aid: the weight-gradient launch over point counts, per variant library (tools/build_variant.py)
for so in "" build/perjob0.so build/perjobmax.so; do
  echo "== ${so:-product}"
  for rs in "64 64" "128 64" "512 64" "1024 64" "2048 64" "4096 64" "8192 64" "20480 64"; do set -- $rs; LUSH_SO=$so R=$1 S=$2 MODES=h,h WHAT=weights REPS=5 python tools/bench_mlp.py 2>&1 | tail -1; done
done
