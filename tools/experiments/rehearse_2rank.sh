mkdir -p gpurun_out/r5j
export COGS_BENCH_REHEARSAL=1
S=$(date +%s)
timeout -k 10 600 python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r5j/bench2.json 2> gpurun_out/r5j/bench2.err
echo "rc $? wall $(( $(date +%s) - S )) s"
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r5j/bench2.json") if l.startswith("{")][0])
print(d["value"], d["n_gpus"], d["ranks_seen"], d["ms_per_step"]); print(json.dumps(d["allgather"], indent=1)); print(d.get("wall_s")); print(d["cfg3"]["value"], d["weak"]["value"])
PY
tail -3 gpurun_out/r5j/bench2.err
