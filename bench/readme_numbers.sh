# Collects the numbers of the README table on one GPU into gpurun_out/readme_nums.txt: bash bench/readme_numbers.sh
o=gpurun_out/readme_nums.txt; : > $o
b() { echo "== $*" >> $o; timeout -k 10 300 python bench.py "$@" 2>/dev/null | tail -1 >> $o; }
b --steps 20 --warmup 5
b --steps 20 --warmup 5 --batch-per-gpu 32768 --eval 0
b --steps 20 --warmup 5 --batch-per-gpu 65536 --eval 0
b --steps 20 --warmup 5 --batch-per-gpu 131072 --eval 0
b --steps 20 --warmup 5 --batch-per-gpu 524288 --eval 0
b --model mnist_cnn --steps 20 --warmup 5
b --model wide_mlp --steps 5 --warmup 2 --eval 0
b --model wide_mlp --steps 5 --warmup 2 --eval 0 --batch-per-gpu 2048
b --steps 20 --warmup 5 --matmul fp32_split --eval 0
b --model xor_mlp --steps 20 --warmup 5 --eval 0 --fused 0
python - <<'PY' >> gpurun_out/readme_nums.txt
import json
for l in open("gpurun_out/readme_nums.txt"):
    if l.startswith("{"):
        d = json.loads(l); print(d["config"]["model"], d["config"]["batch_per_gpu"], round(d["ms_per_step"], 4), round(d["value"]), d["config"].get("test_acc"))
PY
tail -12 gpurun_out/readme_nums.txt
