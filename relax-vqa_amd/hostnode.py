"""Host-side placement of the feeding threads of one rank: which NUMA node its GPU hangs off, which CPUs belong to that node, and
how much pinned staging memory the rank may keep.

The reference's per-video loop (src/main_fragment_layerstack.py:269-296) reads frames on the one thread it runs on; here every rank
of a node has `workers` loader threads and a pinned staging pool (dataset.ClipStager), and eight ranks of them share two sockets:
  * a loader thread that decodes on the far socket and a pinned buffer that lives there cross the inter-socket link twice per byte
    (decode -> pinned -> PCIe root of the GPU): the loader threads of a rank are bound to the CPUs of its GPU's node, and they are
    the ones that allocate the rank's pinned buffers (first touch -> that node);
  * pinned memory is not pageable: the pool a rank keeps for reuse is a share of a per-NODE budget (RELAX_PINNED_POOL_GB, default
    64 GiB for the whole node), divided by the ranks on the node.
Everything here degrades to "do nothing" when the platform does not tell (containers without /sys, numa_node = -1)."""
import os


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def parse_cpulist(text):
    """'0-3,8,10-11' -> {0, 1, 2, 3, 8, 10, 11}"""
    cpus = set()
    for part in (text or "").split(","):
        part = part.strip()
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def pci_address(device_index):
    """'dddd:bb:dd.f' of a HIP device as sysfs names it, or None."""
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        return f"{int(p.pci_domain_id):04x}:{int(p.pci_bus_id):02x}:{int(p.pci_device_id):02x}.0"
    except Exception:          # noqa: BLE001 - no GPU, or a torch build without the PCI fields
        return None


def gpu_numa_node(device_index, sysfs="/sys"):
    """NUMA node of the GPU's PCIe root, or None if the platform does not say (file missing, or -1)."""
    addr = pci_address(device_index)
    if addr is None:
        return None
    text = _read(os.path.join(sysfs, "bus", "pci", "devices", addr, "numa_node"))
    try:
        node = int(text)
    except (TypeError, ValueError):
        return None
    return node if node >= 0 else None


def node_cpus(node, sysfs="/sys"):
    """CPUs of a NUMA node that this process may run on (empty set: unknown)."""
    cpus = parse_cpulist(_read(os.path.join(sysfs, "devices", "system", "node", f"node{node}", "cpulist")))
    try:
        allowed = os.sched_getaffinity(0)
    except (AttributeError, OSError):
        allowed = cpus
    return cpus & set(allowed)


def bind_this_thread(cpus):
    """Restrict the CALLING thread (Linux: sched_setaffinity(0) is per thread) to `cpus`; False if that was not possible."""
    if not cpus:
        return False
    try:
        os.sched_setaffinity(0, cpus)
        return True
    except (AttributeError, OSError, ValueError):
        return False


def ranks_share_gpus():
    """True when this node runs more ranks than it has GPUs (the gloo rehearsal of N ranks on a one-GPU box)."""
    try:
        import torch
        n = torch.cuda.device_count()
    except Exception:          # noqa: BLE001
        return False
    return n > 0 and local_world_size() > n


def loader_cpus(device_index, sysfs="/sys"):
    """The CPU set the loader threads of the rank that drives `device_index` should run on: the CPUs of the GPU's NUMA node, or
    None (leave the threads where the scheduler puts them): RELAX_NUMA_BIND=0, a CPU 'device', no NUMA information - or ranks that
    SHARE a GPU (RELAX_NUMA_BIND unset): binding is for one rank per GPU, where the ranks spread over the sockets as the GPUs do; N
    ranks on one GPU would all pile onto that GPU's socket (measured in the 8-rank rehearsal on a one-GPU box: 591 against 698
    clips/s aggregate at config 4, profiles/r05_rehearsal_*).  RELAX_NUMA_BIND=1 forces the binding."""
    mode = os.environ.get("RELAX_NUMA_BIND", "")
    if mode == "0" or device_index is None or (mode != "1" and ranks_share_gpus()):
        return None
    node = gpu_numa_node(device_index, sysfs)
    if node is None:
        return None
    cpus = node_cpus(node, sysfs)
    return cpus or None


def local_world_size(world=1):
    """Ranks on this node (torchrun exports LOCAL_WORLD_SIZE); falls back to the world size of a single-node job."""
    try:
        return max(int(os.environ.get("LOCAL_WORLD_SIZE", "")), 1)
    except ValueError:
        return max(int(world), 1)


def pinned_pool_budget(world=1):
    """Bytes of pinned staging ONE rank keeps for reuse: the node's budget (RELAX_PINNED_POOL_GB, default 64) over its ranks."""
    try:
        node_gb = float(os.environ.get("RELAX_PINNED_POOL_GB", "64"))
    except ValueError:
        node_gb = 64.0
    return int(node_gb * (1 << 30) / local_world_size(world))
