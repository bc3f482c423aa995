# usage: bash profiles/abf.sh "<bench args>"  — fused single-launch path loop vs one launch per stage, interleaved twice
for round in 1 2; do
  for f in 1 0; do
    CRT_FUSED=$f timeout -k 10 200 python bench.py $1 --no-cpu-baseline 2>>gpurun_out/ab_stderr.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fused=$f', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
  done
done
