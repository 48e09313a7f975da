"""Host placement of one rank per GPU (replicas only, one process per GPU - the reference's global LoopTimer,
include/eqvio/LoopTimer.h:95, makes process-per-filter the compatible mode anyway).

A filter's frame boundary goes through the host (doorbell -> results -> next launch, ~13 us of a 90 us frame): on a two-socket 8-GPU node a
rank whose spinning host thread sits on the other socket than its GPU pays the cross-socket hop on every doorbell poll and every launch, and
the per-rank rates spread. Before a rank creates its HIP context (and with it the pinned doorbell / result packets, which are placed by first
touch), it is pinned to a small block of physical cores of the NUMA node its GPU hangs off:

    node of a GPU      /sys/bus/pci/devices/<domain:bus:dev.fn>/numa_node
    cores of a node    /sys/devices/system/node/node<k>/cpulist
    SMT siblings       /sys/devices/system/cpu/cpu<c>/topology/thread_siblings_list   (one thread per physical core is used)

Everything takes a `sysfs` root so that the mapping logic is testable with a faked tree (tests/test_placement.py). Nothing here touches a GPU."""
import os


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    out = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def numa_node_of_pci(bus_id, sysfs="/sys"):
    """NUMA node of the PCI function `bus_id` ('0000:c1:00.0'); -1 when the platform does not say (single socket, or a VM)."""
    t = _read(os.path.join(sysfs, "bus", "pci", "devices", bus_id.lower(), "numa_node"))
    try:
        return int(t) if t is not None else -1
    except ValueError:
        return -1


def cpus_of_node(node, sysfs="/sys"):
    t = _read(os.path.join(sysfs, "devices", "system", "node", "node%d" % node, "cpulist"))
    return parse_cpulist(t) if t else []


def physical_cores(cpus, sysfs="/sys"):
    """One logical CPU per physical core (the lowest-numbered sibling), in ascending order."""
    seen, out = set(), []
    for c in sorted(cpus):
        t = _read(os.path.join(sysfs, "devices", "system", "cpu", "cpu%d" % c, "topology", "thread_siblings_list"))
        sib = tuple(parse_cpulist(t)) if t else (c,)
        key = min(sib)
        if key in seen:
            continue
        seen.add(key)
        out.append(c)
    return out


def plan(gpu_bus_ids, allowed, cores_per_rank=4, sysfs="/sys"):
    """Core block of every local rank. gpu_bus_ids[r] = PCI bus id of rank r's GPU (several ranks may name the same GPU: the one-device rehearsal),
    allowed = the logical CPUs this job may run on. Ranks whose GPUs share a NUMA node get consecutive, disjoint blocks of that node's physical cores;
    a node that cannot give every one of its ranks a full block gives equal shares (at least one core), and a GPU without NUMA information takes its
    share of ALL allowed cores. Returns a list of dicts: rank, bus_id, numa_node, cpus."""
    allowed = sorted(set(allowed))
    nodes = [numa_node_of_pci(b, sysfs) for b in gpu_bus_ids]
    out = []
    for r, (bus, node) in enumerate(zip(gpu_bus_ids, nodes)):
        peers = [q for q in range(len(gpu_bus_ids)) if nodes[q] == node]
        pool = [c for c in (cpus_of_node(node, sysfs) if node >= 0 else allowed) if c in allowed] or allowed
        cores = physical_cores(pool, sysfs) or pool
        share = max(1, min(cores_per_rank, len(cores) // len(peers)))
        k = peers.index(r)
        mine = cores[k * share:(k + 1) * share] or [cores[k % len(cores)]]
        out.append({"rank": r, "bus_id": bus, "numa_node": node, "cpus": mine})
    return out


def gpu_bus_id(device):
    """PCI bus id of HIP device `device` as sysfs spells it, through torch (no context is created by the property query on ROCm builds that cache it;
    where it would be, the id comes from the environment instead: EQVIO_GPU_BUS_IDS=id0,id1,...)."""
    env = os.environ.get("EQVIO_GPU_BUS_IDS")
    if env:
        ids = env.split(",")
        return ids[device % len(ids)]
    import torch

    p = torch.cuda.get_device_properties(device)
    dom = getattr(p, "pci_domain_id", 0)
    return "%04x:%02x:%02x.0" % (dom, p.pci_bus_id, p.pci_device_id)


def pin_rank(local_rank, world_size, devices=None, cores_per_rank=4, sysfs="/sys"):
    """Pin THIS process to its block (os.sched_setaffinity) and return its plan entry. devices[r] = HIP device of local rank r (default: r).
    Call before the first HIP call of the process."""
    devices = list(range(world_size)) if devices is None else devices
    try:
        bus = [gpu_bus_id(d) for d in devices]
    except Exception as e:  # no torch / no device: nothing to pin against
        return {"rank": local_rank, "bus_id": None, "numa_node": -1, "cpus": sorted(os.sched_getaffinity(0)), "pinned": False, "why": repr(e)}
    entry = dict(plan(bus, os.sched_getaffinity(0), cores_per_rank, sysfs)[local_rank])
    try:
        # Every thread the process has by now, not only the caller: the bus-id query above may have started the HIP runtime (it does on builds that do not cache the
        # property), and the runtime's helper threads were created with the old mask - sched_setaffinity(0, ...) alone pins the calling thread only.
        # Threads created from here on inherit the caller's mask.
        cpus = set(entry["cpus"])
        tids = []
        try:
            tids = [int(t) for t in os.listdir("/proc/self/task")]
        except OSError:
            pass
        moved = 0
        for tid in tids:
            try:
                os.sched_setaffinity(tid, cpus)
                moved += 1
            except OSError:  # a thread that has just exited
                pass
        os.sched_setaffinity(0, cpus)
        entry["pinned"], entry["threads_pinned"] = True, max(moved, 1)
    except OSError as e:
        entry["pinned"], entry["why"] = False, repr(e)
    return entry
