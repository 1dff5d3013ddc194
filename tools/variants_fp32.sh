for v in 0 1 2 0; do
  GDDIM_IGEMM_VARIANT=$v python bench.py --no-cpu-baseline --no-alt --steps 2 --warmup 1 2>/dev/null | tail -1 > /tmp/v.json
  python - <<PY
import json
d=json.loads(open("/tmp/v.json").read())
print("variant $v: %.2f img/s" % d["value"], [(k["name"][-22:], round(k["ms"]), round(k["tflops"] or 0,1)) for k in d["kernel_breakdown"][:4]])
PY
done
GDDIM_IGEMM_SMALL4=1 python bench.py --no-cpu-baseline --no-alt --steps 2 --warmup 1 2>/dev/null | tail -1 > /tmp/v.json
python - <<PY
import json
d=json.loads(open("/tmp/v.json").read())
print("small4: %.2f img/s" % d["value"], [(k["name"][-22:], round(k["ms"]), round(k["tflops"] or 0,1)) for k in d["kernel_breakdown"][:4]])
PY
