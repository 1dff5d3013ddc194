#!/bin/bash
# Socket power, shader clock and temperature while the sampler runs in each precision mode (fp32 | split6 | bf16x3): rocm-smi sampled twice a
# second beside `bench.py --precision M` (8 timed steps, no CPU leg, no secondary modes).  The MI355X socket is capped at 1.4 kW; a mode whose
# kernels draw more power per unit time runs -- and makes its neighbours run -- at a lower shader clock.
#   bash tools/power_modes.sh [outdir]      (outdir default gpurun_out/power)
out=${1:-gpurun_out/power}
mkdir -p "$out"
for mode in fp32 split6 bf16x3; do
    ( while true; do
          rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Socket Graphics Package Power|sclk clock level|junction" | tr '\n' ';'
          echo
          sleep 0.5
      done ) > "$out/smi_$mode.log" &
    smi=$!
    python bench.py --precision "$mode" --steps 8 --warmup 3 --no-cpu-baseline --no-profile --no-alt > "$out/bench_$mode.json" 2> "$out/bench_$mode.err"
    kill $smi
    wait $smi 2>/dev/null
done
python - "$out" <<'EOF'
import json, re, sys, statistics
out = sys.argv[1]
res = {}
for mode in ("fp32", "split6", "bf16x3"):
    pw, ck = [], []
    for line in open("%s/smi_%s.log" % (out, mode)):
        p = re.search(r"Package Power \(W\): ([0-9.]+)", line)
        c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", line)
        if p and c and float(p.group(1)) > 600:  # samples inside the sampler calls
            pw.append(float(p.group(1))); ck.append(float(c.group(1)))
    try:
        line = json.loads(open("%s/bench_%s.json" % (out, mode)).read().strip().splitlines()[-1])
    except Exception:
        line = {}
    res[mode] = {"samples": len(pw), "power_w_median": statistics.median(pw) if pw else None, "power_w_max": max(pw) if pw else None,
                 "sclk_mhz_median": statistics.median(ck) if ck else None, "images_per_s": line.get("value"), "ms_per_step_median": line.get("ms_per_step_median")}
print(json.dumps(res, indent=1))
open("%s/power_modes.json" % out, "w").write(json.dumps(res, indent=1) + "\n")
EOF
