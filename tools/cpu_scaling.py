"""How the CPU baseline of bench.py scales over processes x threads on this box (the GPU boxes show 256 logical CPUs):
python tools/cpu_scaling.py [budget_s]"""
import json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gddim_amd.configs import cld_accr_dcifar10
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
cfg = cld_accr_dcifar10.get_config()
if __name__ == "__main__":
    print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "granted", bench.effective_cpus())
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
        try:
            print(f, open(f).read().strip())
        except OSError:
            pass
    print(subprocess.run("lscpu | grep -E 'Model name|Socket|Core|Thread|NUMA node\\(s\\)|^CPU\\(s\\)'; free -g | head -2; uptime", shell=True, capture_output=True, text=True).stdout)
    for P, T in ((1, 16), (1, 32), (4, 16), (8, 16), (16, 16), (16, 8), (32, 8), (64, 4)):
        r = bench.cpu_baseline(cfg, 50, 16, 20, T, budget, P)
        print(P, T, json.dumps({k: r.get(k) for k in ("value", "seconds", "evaluations_per_process", "per_process_seconds")}), flush=True)
