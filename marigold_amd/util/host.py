"""Host facts for the CPU-side legs (tests' oracle runs, bench.py's cpu_baseline)."""
import os


def usable_cores() -> int:
    """Cores this process may actually use: scheduler affinity capped by the cgroup CPU quota.
    ``os.cpu_count()`` reports the machine, not the container - sizing an OpenMP pool by it inside
    a quota-limited container oversubscribes the cores and makes small torch ops crawl."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:            # cgroup v2
            q, p = f.read().split()[:2]
            if q != "max":
                quota = int(q) / int(p)
    except (OSError, ValueError):
        try:                                                   # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                p = int(f.read())
            if q > 0 and p > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota)))
    return max(1, n)


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def build_fingerprint():
    """sha256 over everything that decides which kernels a map launches and what they do: the library's sources (csrc/*.hip, *.h,
    *.inc, Makefile), the program builder (engine.py, ops.py, weights.py) and the tuning table.  bench.py prints it and compares it
    with the one recorded beside the PMC traffic figures (profiles/r*_pmc_hbm_traffic.json: ``traffic_stale``) - the GPU box has
    no .git, a content hash travels."""
    import glob
    import hashlib
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "csrc", "*.hip")) + glob.glob(os.path.join(root, "csrc", "*.h")) +
                   glob.glob(os.path.join(root, "csrc", "*.inc")) + [os.path.join(root, "csrc", "Makefile")] +
                   [os.path.join(root, n) for n in ("engine.py", "ops.py", "weights.py")] +
                   glob.glob(os.path.join(root, "tuning", "*.json")))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.relpath(f, root).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
