"""Host placement of the ranks (eqvio_amd/placement.py): the mapping GPU -> NUMA node -> block of physical cores, on faked sysfs trees.
No GPU, no torch: the PCI bus ids are given."""
import os

import pytest

from eqvio_amd.placement import cpus_of_node, numa_node_of_pci, parse_cpulist, physical_cores, pin_rank, plan


def fake_sysfs(root, nodes, smt=2, gpus=None):
    """nodes: {node: [physical core count]}; logical CPUs are numbered like a two-socket EPYC box: first the first threads of every core (node 0, node 1, ...),
    then the second threads in the same order. gpus: {bus_id: node}."""
    ncores = sum(nodes.values())
    first, cpu = {}, 0
    for node, cnt in sorted(nodes.items()):
        first[node] = list(range(cpu, cpu + cnt))
        cpu += cnt
    for node, cores in first.items():
        logical = sorted(c + t * ncores for c in cores for t in range(smt))
        d = root / "devices" / "system" / "node" / f"node{node}"
        d.mkdir(parents=True)
        # a cpulist the way the kernel prints it: ranges
        lo, hi = cores[0], cores[-1]
        (d / "cpulist").write_text(",".join(f"{lo + t * ncores}-{hi + t * ncores}" for t in range(smt)) + "\n")
        for c in logical:
            t = root / "devices" / "system" / "cpu" / f"cpu{c}" / "topology"
            t.mkdir(parents=True)
            base = c % ncores
            (t / "thread_siblings_list").write_text(",".join(str(base + k * ncores) for k in range(smt)) + "\n")
    for bus, node in (gpus or {}).items():
        d = root / "bus" / "pci" / "devices" / bus
        d.mkdir(parents=True)
        (d / "numa_node").write_text(f"{node}\n")
    return str(root)


def test_cpulist_forms():
    assert parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert parse_cpulist("") == [] and parse_cpulist("5") == [5]


def test_two_socket_eight_gpu_node(tmp_path):
    """8 GPUs, 4 per socket, 2 x 64 cores with SMT: every rank gets 4 physical cores of its GPU's node, disjoint from every other rank's, no SMT sibling pairs."""
    gpus = {f"0000:{b:02x}:00.0": (0 if k < 4 else 1) for k, b in enumerate((0x05, 0x15, 0x25, 0x35, 0x85, 0x95, 0xa5, 0xb5))}
    sysfs = fake_sysfs(tmp_path, {0: 64, 1: 64}, smt=2, gpus=gpus)
    assert numa_node_of_pci("0000:85:00.0", sysfs) == 1 and numa_node_of_pci("0000:05:00.0", sysfs) == 0
    assert cpus_of_node(1, sysfs)[:3] == [64, 65, 66] and len(cpus_of_node(1, sysfs)) == 128
    assert physical_cores(cpus_of_node(1, sysfs), sysfs) == list(range(64, 128))
    pl = plan(list(gpus), range(256), cores_per_rank=4, sysfs=sysfs)
    assert [e["numa_node"] for e in pl] == [0, 0, 0, 0, 1, 1, 1, 1]
    seen = set()
    for e in pl:
        assert len(e["cpus"]) == 4 and not (set(e["cpus"]) & seen)
        seen |= set(e["cpus"])
        lo, hi = (0, 64) if e["numa_node"] == 0 else (64, 128)
        assert all(lo <= c < hi for c in e["cpus"])  # first threads of cores of the right socket only
    assert pl[0]["cpus"] == [0, 1, 2, 3] and pl[3]["cpus"] == [12, 13, 14, 15] and pl[4]["cpus"] == [64, 65, 66, 67]


def test_one_device_rehearsal_and_restricted_cpuset(tmp_path):
    """EQVIO_BENCH_ONE_DEVICE: eight ranks on ONE GPU share that GPU's node and still get disjoint blocks; a cgroup that leaves only a few CPUs gives equal shares."""
    sysfs = fake_sysfs(tmp_path, {0: 16, 1: 16}, smt=2, gpus={"0000:c1:00.0": 1})
    pl = plan(["0000:c1:00.0"] * 8, range(64), cores_per_rank=4, sysfs=sysfs)
    blocks = [tuple(e["cpus"]) for e in pl]
    assert all(e["numa_node"] == 1 for e in pl) and len(set(blocks)) == 8
    assert all(len(b) == 2 for b in blocks)  # 16 physical cores / 8 ranks
    assert sorted(c for b in blocks for c in b) == list(range(16, 32))
    few = plan(["0000:c1:00.0"] * 2, [16, 17, 48, 49, 0], cores_per_rank=4, sysfs=sysfs)  # CPUs 48 / 49 are the SMT siblings of 16 / 17; CPU 0 is on the other node
    assert [e["cpus"] for e in few] == [[16], [17]]


def test_no_numa_information(tmp_path):
    """A VM / single socket: numa_node = -1 (or no sysfs entry at all): the ranks share out the allowed CPUs."""
    sysfs = fake_sysfs(tmp_path, {0: 8}, smt=1, gpus={"0000:03:00.0": -1})
    pl = plan(["0000:03:00.0", "0000:04:00.0"], range(8), cores_per_rank=4, sysfs=sysfs)  # the second GPU has no entry
    assert [e["numa_node"] for e in pl] == [-1, -1]
    assert pl[0]["cpus"] == [0, 1, 2, 3] and pl[1]["cpus"] == [4, 5, 6, 7]


def test_pin_rank_pins_this_process(tmp_path, monkeypatch):
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 2:
        pytest.skip("needs two CPUs")
    n = max(allowed) + 1
    sysfs = fake_sysfs(tmp_path, {0: n}, smt=1, gpus={"0000:03:00.0": 0, "0000:04:00.0": 0})
    monkeypatch.setenv("EQVIO_GPU_BUS_IDS", "0000:03:00.0,0000:04:00.0")
    try:
        e = pin_rank(1, 2, cores_per_rank=1, sysfs=sysfs)
        assert e["pinned"] and e["numa_node"] == 0 and e["cpus"] == [allowed[1]]
        assert sorted(os.sched_getaffinity(0)) == [allowed[1]]
    finally:
        os.sched_setaffinity(0, allowed)


def test_pin_rank_pins_the_threads_that_exist_already(tmp_path, monkeypatch):
    """the bus-id query may start the HIP runtime, whose helper threads keep the old mask under sched_setaffinity(0, ...) alone: every thread of
    the process is moved (a thread started before the call stands in for the runtime's)."""
    import threading

    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 2:
        pytest.skip("needs two CPUs")
    n = max(allowed) + 1
    sysfs = fake_sysfs(tmp_path, {0: n}, smt=1, gpus={"0000:03:00.0": 0})
    monkeypatch.setenv("EQVIO_GPU_BUS_IDS", "0000:03:00.0")
    stop, tid, seen = threading.Event(), [], []

    def helper():
        tid.append(threading.get_native_id())
        stop.wait(10)
        seen.append(sorted(os.sched_getaffinity(0)))  # (0 = the calling thread)

    t = threading.Thread(target=helper)
    t.start()
    while not tid:
        pass
    try:
        e = pin_rank(0, 1, cores_per_rank=1, sysfs=sysfs)
        assert e["pinned"] and e["threads_pinned"] >= 2
        assert sorted(os.sched_getaffinity(tid[0])) == e["cpus"]
    finally:
        stop.set()
        t.join()
        for x in os.listdir("/proc/self/task"):
            try:
                os.sched_setaffinity(int(x), allowed)
            except OSError:
                pass
    assert seen == [e["cpus"]]
