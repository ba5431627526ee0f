bash tools/profile_round.sh r01 > /dev/null 2>&1
bash tools/profile_bf16.sh r01 > /dev/null 2>&1
ls gpurun_out/prof | head -40
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/prof/r01_*bench*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["achieved"], d.get("bf16_storage_mode",{}).get("value"))
    except Exception as e: print(f, "ERR", e)
PY
