for v in 0 1 0 1; do
  GDDIM_X3_VARIANT=$v python bench.py --precision bf16x3 --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | tail -1 > /tmp/v.json
  python - <<PY
import json
d=json.loads(open("/tmp/v.json").read())
print("x3 variant $v: %.2f img/s" % d["value"], [(k["name"][-24:], round(k["ms"]), round(k["tflops"] or 0,1)) for k in d["kernel_breakdown"][:3]])
PY
done
