"""Per-device NUMA placement (efx_numa_*, include/efx.h) against a fake sysfs tree: which node a PCI device sits on, that
node's CPUs, and binding the calling thread -- what efx_multi_create does for each device's worker thread and bench.py for
each rank, so that at 8 ranks the ingest leg's staging memory and copy threads stay on the device's own socket.  Host only."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_tree(root, nodes):
    """nodes: {node: (cpulist, [pci ids])}"""
    for node, (cpulist, devs) in nodes.items():
        d = os.path.join(root, "sys", "devices", "system", "node", f"node{node}")
        os.makedirs(d)
        with open(os.path.join(d, "cpulist"), "w") as f:
            f.write(cpulist + "\n")
        for dev in devs:
            p = os.path.join(root, "sys", "bus", "pci", "devices", dev)
            os.makedirs(p)
            with open(os.path.join(p, "numa_node"), "w") as f:
                f.write(f"{node}\n")


def run_py(root, code):
    env = dict(os.environ, EFX_SYSFS_ROOT=str(root), PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], capture_output=True, text=True, env=env, timeout=120)
    assert p.returncode == 0, p.stderr
    return p.stdout.strip()


def test_node_and_cpus_from_sysfs(tmp_path):
    avail = sorted(os.sched_getaffinity(0))
    # node 0 = the first half of the CPUs this process may use (as ranges), node 1 = CPUs it does not have
    half = avail[: max(1, len(avail) // 2)]
    make_tree(tmp_path, {0: (",".join(str(c) for c in half), ["0000:05:00.0", "0000:15:00.0"]),
                         1: ("1000-1003", ["0000:c1:00.0"]),
                         2: (f"{avail[0]}-{avail[0]}", ["0001:e5:00.0"])})
    out = run_py(tmp_path, """
        import os, espflix_amd as efx
        print(efx.numa_node_of_pci("0000:05:00.0"), efx.numa_node_of_pci("0000:C1:00.0"), efx.numa_node_of_pci("c1:00.0"),
              efx.numa_node_of_pci("0001:e5:00.0"), efx.numa_node_of_pci("0000:99:00.0"))
        print(efx.numa_cpus_of_node(1), efx.numa_cpus_of_node(7))
        before = sorted(os.sched_getaffinity(0))
        print(efx.numa_bind_thread(1), sorted(os.sched_getaffinity(0)) == before)      # none of node 1's CPUs are ours: left alone
        print(efx.numa_bind_thread(-1), sorted(os.sched_getaffinity(0)) == before)     # unknown node: left alone
        n = efx.numa_bind_thread(0)
        print(n, sorted(os.sched_getaffinity(0)))
        """).splitlines()
    assert out[0] == "0 1 1 2 -1"
    assert out[1] == "[1000, 1001, 1002, 1003] []"
    assert out[2] == "0 True" and out[3] == "0 True"
    assert out[4] == f"{len(half)} {half}"


def test_bench_rank_binding_helper(tmp_path):
    """espflix_amd.dist.bind_rank_to_device_node: what a bench.py rank does with its GPU's PCI bus id."""
    avail = sorted(os.sched_getaffinity(0))
    make_tree(tmp_path, {3: (f"{avail[-1]}", ["0000:75:00.0"])})
    out = run_py(tmp_path, """
        import os
        from espflix_amd import dist
        print(dist.bind_rank_to_device_node("0000:75:00.0"), sorted(os.sched_getaffinity(0)))
        print(dist.bind_rank_to_device_node("0000:76:00.0"))
        """).splitlines()
    assert out[0] == f"{{'node': 3, 'cpus_bound': 1}} [{avail[-1]}]"
    assert out[1] == "{'node': -1, 'cpus_bound': 0}"


def test_scale_harness_dry_run_eight_devices(tmp_path):
    """tools/efx_scale --dry-run: the 8-device job's partition (stream k on device floor(k * 8 / 8192))
    and the NUMA node each device's host thread would be bound to, without touching a device."""
    exe = os.path.join(ROOT, "tools", "efx_scale")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", ROOT, "scale"], check=True, capture_output=True)
    devs = [f"0000:{0x05 + 0x10 * i:02x}:00.0" for i in range(8)]
    make_tree(tmp_path, {0: ("0-47", devs[:4]), 1: ("48-95", devs[4:])})
    p = subprocess.run([exe, "--dry-run", "1", "--devices", "8", "--streams", "1024", "--pci", ",".join(devs)], capture_output=True, text=True,
                       env=dict(os.environ, EFX_SYSFS_ROOT=str(tmp_path)), timeout=60)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = p.stdout.splitlines()
    assert lines[-1] == "DRY_RUN devices=8 streams_total=8192 covered=8192"
    for r in range(8):
        assert lines[r].startswith(f"device {r}: streams [{r * 1024}, {(r + 1) * 1024}) = 1024,")
        assert f"numa node {0 if r < 4 else 1}, 48 cpus" in lines[r]
    # an uneven job: 3 devices
    p = subprocess.run([exe, "--dry-run", "1", "--devices", "3", "--streams", "1000"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0 and p.stdout.splitlines()[-1] == "DRY_RUN devices=3 streams_total=3000 covered=3000"
