for cfg in "384 384" "256 384" "384 128" "256 128" "384 384"; do set -- $cfg
  GDDIM_IGEMM_BIGMIN=$1 GDDIM_IGEMM_MIDMIN=$2 python bench.py --no-cpu-baseline --no-alt --steps 2 --warmup 1 2>/dev/null | tail -1 > /tmp/v.json
  python - <<PY
import json
d=json.loads(open("/tmp/v.json").read())
print("big>=$1 mid>=$2: %.2f img/s" % d["value"], [(k["name"][-24:], round(k["ms"]), round(k["tflops"] or 0,1), k["launches"]) for k in d["kernel_breakdown"][:4] if "igemm" in k["name"]])
PY
done
