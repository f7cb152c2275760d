"""Where the host side of a launch lives, and whether it explains the leases on which the small launches take three times as long and the
streaming kernels are 5-12 % slower inside the step (profiles/r05/run20_*, run22_*: PCI 0a:00, 5a:00, 5d:00 against 23:00, 0d:00, a7:00).

Prints the NUMA layout of the box (the GPU's node and local CPUs from sysfs, the nodes' CPU lists, this process's affinity), then runs
`bench.py --no-cpu-baseline --no-extras` and a loop of 400 small launches (km_homography_chain_fwd, B = 256: one 64-thread workgroup per
four images) under: the default placement, the process bound to the GPU's local CPUs, bound to the CPUs of the farthest other node,
and HIP_FORCE_DEV_KERNARG = 0 / 1 (kernel arguments in host / device memory).

    python profiles/numa_ab.py            (device_run.sh: py:profiles/numa_ab.py)
"""
import glob, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def micro():
    sys.path.insert(0, ROOT)
    import torch
    import bench
    from kornia_amd import _native as N
    lib = N.lib(); dev = torch.device("cuda"); stream = N.stream_ptr(dev)
    B, S = 256, 512
    M = bench.flagship_homographies(B, S, S, torch.Generator().manual_seed(0)).to(dev)
    m = torch.empty(B, 9, device=dev)
    def f():
        lib.km_homography_chain_fwd(M.data_ptr(), 3, None, m.data_ptr(), B, S, S, S, S, 0, stream)
    ts = sorted(bench.event_time_ms(f, 400) for _ in range(5))
    pr = torch.cuda.get_device_properties(0)
    print(json.dumps({"small_launch_us": [round(t * 1e3, 2) for t in ts], "pci": f"{pr.pci_bus_id:02x}"}))


def parse_cpulist(s):
    out = set()
    for part in s.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out


def run(label, env_extra, cpus):
    env = dict(os.environ); env.update(env_extra)
    pre = (lambda: os.sched_setaffinity(0, cpus)) if cpus else None
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extras"], env=env, preexec_fn=pre, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    d = json.loads(line[-1]) if line else {}
    r2 = subprocess.run([sys.executable, os.path.abspath(__file__), "--micro"], env=env, preexec_fn=pre, capture_output=True, text=True)
    line2 = [l for l in r2.stdout.splitlines() if l.startswith("{")]
    d2 = json.loads(line2[-1]) if line2 else {"err": r2.stderr[-300:]}
    ops = {k.replace("km_", ""): v["ms"] for k, v in d.get("ops", {}).items()}
    print(f"{label:34s} step {d.get('ms_per_step')} ms   ops {ops}   small launches (400 in a row, us each, 5 repeats) {d2.get('small_launch_us', d2)}", flush=True)
    return d


if __name__ == "__main__":
    if "--micro" in sys.argv:
        micro(); sys.exit(0)
    allowed = os.sched_getaffinity(0)
    nodes = {}
    for p in sorted(glob.glob("/sys/devices/system/node/node*/cpulist")):
        nodes[int(p.split("node")[-1].split("/")[0])] = parse_cpulist(open(p).read())
    print("nodes:", {n: f"{min(c)}-{max(c)} ({len(c)} cpus, {len(c & allowed)} allowed)" for n, c in nodes.items() if c})
    print("affinity of this process:", len(allowed), "cpus", sorted(allowed)[:4], "...", sorted(allowed)[-4:])
    for p in sorted(glob.glob("/sys/devices/system/node/node*/distance")):
        print(p.split("/")[-2], "distance", open(p).read().strip())
    d0 = run("default", {}, None)
    pci = d0.get("clocks", {}).get("before", {}).get("pci")
    near = None; gnode = None
    if pci:
        base = os.path.join("/sys/bus/pci/devices", pci)
        try:
            gnode = int(open(os.path.join(base, "numa_node")).read())
            near = parse_cpulist(open(os.path.join(base, "local_cpulist")).read())
        except OSError as e:
            print("no NUMA files for", pci, e)
    print("GPU", pci, "numa_node", gnode, "local cpus", (f"{min(near)}-{max(near)} ({len(near)}, {len(near & allowed)} allowed)" if near else None), flush=True)
    if near and (near & allowed):
        run("bound to the GPU's local CPUs", {}, near & allowed)
    far = None
    if gnode is not None and gnode >= 0 and len(nodes) > 1:
        try:
            dist = [int(x) for x in open(f"/sys/devices/system/node/node{gnode}/distance").read().split()]
            order = sorted((n for n in nodes if n != gnode and (nodes[n] & allowed)), key=lambda n: -dist[n])
            if order:
                far = nodes[order[0]] & allowed
                run(f"bound to node {order[0]} (farthest allowed)", {}, far)
        except (OSError, IndexError) as e:
            print("no distance table:", e)
    run("HIP_FORCE_DEV_KERNARG=0", {"HIP_FORCE_DEV_KERNARG": "0"}, None)
    run("HIP_FORCE_DEV_KERNARG=1", {"HIP_FORCE_DEV_KERNARG": "1"}, None)
    run("default again", {}, None)
