"""Cluster definition and the per-task server.

Surface of /root/reference/example.py:108-141: ``ClusterSpec({"ps": [...], "worker": [...]})``,
``Server(cluster, job_name=..., task_index=...)``, ``server.join()`` for parameter-server
tasks, ``server.target`` and ``replica_device_setter(worker_device=..., cluster=...)``.

MI355X mapping (one node, up to 8 GPUs, one process per task):

* worker task *i* computes on GPU ``i % ngpus`` (CPU when there is no GPU);
* a ps task owns a share of the variables.  ``replica_device_setter`` decides which:
  whole variables round-robin in creation order (the TF default, kept for compatibility),
  byte-balanced greedy, or every variable split 1/N across all owners (``flat``), which is
  what a ≥1B-parameter model wants so that all HBMs and all xGMI links share the load;
* bulk tensors move over RCCL (``torch.distributed`` "nccl" backend) between GPUs - gloo
  between CPU test processes - while rendezvous, request queues, ``global_step`` and
  staleness clocks use the TCP control plane served at each ps task's ``host:port``.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple, Union


class ClusterSpec:
    """Job name -> ordered list of task addresses (``host:port``); task index = list position."""

    def __init__(self, cluster: Union[Dict[str, Sequence[str]], "ClusterSpec"]):
        if isinstance(cluster, ClusterSpec):
            self._jobs = {k: list(v) for k, v in cluster._jobs.items()}
        elif isinstance(cluster, dict):
            self._jobs = {}
            for job, tasks in cluster.items():
                if isinstance(tasks, dict):
                    n = max(tasks) + 1 if tasks else 0
                    lst = [tasks.get(i, "") for i in range(n)]
                else:
                    lst = [t.strip() for t in tasks]
                self._jobs[str(job)] = lst
        else:
            raise TypeError("cluster must be a dict or a ClusterSpec")

    @property
    def jobs(self) -> List[str]:
        return sorted(self._jobs)

    def num_tasks(self, job_name: str) -> int:
        if job_name not in self._jobs:
            raise ValueError("No such job in cluster: %r" % job_name)
        return len(self._jobs[job_name])

    def task_indices(self, job_name: str) -> List[int]:
        return list(range(self.num_tasks(job_name)))

    def task_address(self, job_name: str, task_index: int) -> str:
        tasks = self.job_tasks(job_name)
        if not 0 <= task_index < len(tasks):
            raise ValueError("No task with index %r in job %r" % (task_index, job_name))
        return tasks[task_index]

    def job_tasks(self, job_name: str) -> List[str]:
        if job_name not in self._jobs:
            raise ValueError("No such job in cluster: %r" % job_name)
        return list(self._jobs[job_name])

    def as_dict(self) -> Dict[str, List[str]]:
        return {k: list(v) for k, v in self._jobs.items()}

    def __bool__(self):
        return bool(self._jobs)

    def __eq__(self, other):
        return isinstance(other, ClusterSpec) and self._jobs == other._jobs

    def __repr__(self):
        return "ClusterSpec(%r)" % self.as_dict()


def split_host_port(addr: str) -> Tuple[str, int]:
    host, _, port = addr.rpartition(":")
    if not host or not port.isdigit():
        raise ValueError("task address must be host:port, got %r" % addr)
    return host, int(port)


# --------------------------------------------------------------------------- placement
class PlacementStrategy:
    """Maps variables to parameter-server owners."""

    name = "base"

    def assign(self, variables: Sequence[Tuple[str, int]], num_ps: int) -> Dict[str, int]:
        """``variables``: (name, nbytes) in creation order -> {name: ps task index}."""
        raise NotImplementedError


class RoundRobinPlacement(PlacementStrategy):
    """TF's default ``_RoundRobinStrategy``: k-th created variable -> ps task k mod N, whole tensors."""

    name = "round_robin"

    def assign(self, variables, num_ps):
        return {name: i % max(num_ps, 1) for i, (name, _) in enumerate(variables)}


class ByteBalancedPlacement(PlacementStrategy):
    """``GreedyLoadBalancingStrategy`` with a byte-size load function: next variable -> least loaded ps."""

    name = "byte_balanced"

    def assign(self, variables, num_ps):
        loads = [0] * max(num_ps, 1)
        out = {}
        for name, nbytes in variables:
            k = min(range(len(loads)), key=lambda i: (loads[i], i))
            out[name] = k
            loads[k] += nbytes
        return out


class FlatShardPlacement(PlacementStrategy):
    """Every variable is split evenly across all owners (ranges of the flat buffer); ``assign`` reports
    owner -1 = "all" and the runtime uses ``VariableStore.shard_range``."""

    name = "flat"

    def assign(self, variables, num_ps):
        return {name: -1 for name, _ in variables}


PLACEMENTS = {c.name: c for c in (RoundRobinPlacement, ByteBalancedPlacement, FlatShardPlacement)}


class DeviceSetter:
    """Result of ``replica_device_setter``: callable device function + the placement policy.

    ``setter(op_kind)`` returns the device string TF would have used: variables go to
    ``/job:ps/task:k`` (k by the strategy), everything else to the worker device.
    """

    def __init__(self, worker_device: str, cluster: Optional[ClusterSpec], ps_tasks: int,
                 strategy: PlacementStrategy):
        self.worker_device = worker_device
        self.cluster = cluster
        self.ps_tasks = ps_tasks
        self.strategy = strategy
        self._next = 0
        self._assigned: Dict[str, int] = {}

    def __call__(self, op_kind: str = "op", name: Optional[str] = None, nbytes: int = 0) -> str:
        if op_kind in ("Variable", "VariableV2", "VarHandleOp", "variable") and self.ps_tasks > 0:
            if name is not None and name in self._assigned:
                k = self._assigned[name]
            else:
                k = self._next % self.ps_tasks
                self._next += 1
                if name is not None:
                    self._assigned[name] = k
            return "/job:ps/task:%d" % k
        return self.worker_device

    def place_variables(self, variables: Sequence[Tuple[str, int]]) -> Dict[str, int]:
        """Owner (ps task index) for each (name, nbytes), in creation order."""
        return self.strategy.assign(list(variables), self.ps_tasks)


def replica_device_setter(ps_tasks: int = 0, ps_device: str = "/job:ps", worker_device: str = "/job:worker",
                          cluster: Optional[Union[ClusterSpec, dict]] = None, ps_strategy=None,
                          merge_devices: bool = True) -> Optional[DeviceSetter]:
    """``tf.train.replica_device_setter`` (/root/reference/example.py:136-139)."""
    spec = None
    if cluster is not None:
        spec = cluster if isinstance(cluster, ClusterSpec) else ClusterSpec(cluster)
        ps_tasks = spec.num_tasks("ps") if "ps" in spec.jobs else 0
    if ps_strategy is None:
        ps_strategy = RoundRobinPlacement()
    elif isinstance(ps_strategy, str):
        ps_strategy = PLACEMENTS[ps_strategy]()
    return DeviceSetter(worker_device, spec, ps_tasks, ps_strategy)


# --------------------------------------------------------------------------- device plan
def plan_devices(num_ps: int, num_workers: int, n_gpus: int, shared: bool = False) -> dict:
    """Which device every task of a one-node cluster computes on.

    ``n_gpus == 0``: everything on the CPU.  Otherwise worker ``i`` gets GPU ``i`` and ps task ``j`` GPU
    ``num_workers + j``: RCCL accepts one rank per GPU, so no two tasks of an RCCL cluster may share one.  If only
    the ps tasks do not fit (1 ps + 8 workers on 8 GPUs), they get no GPU at all (``None``): they serve the control
    plane only and the variables are owned by service threads inside the workers (``colocated``), sharded across the
    workers' HBMs.  If the workers alone do not fit (the reference's default local run, 1 ps + 2 workers, on a
    one-GPU machine) the plan is ``oversubscribed``: workers wrap around the GPUs, owners are colocated and exchange
    data through hipIpc peer regions - which work between processes on one device - and any strategy that needs a
    collective group falls back to gloo (``parallel.comm.pick_backend``), because RCCL cannot serve two ranks on one
    GPU.  ``shared=True`` (``DTF_COMM_BACKEND=gloo``: several ranks time-sharing one GPU on a test box, ps tasks
    included) wraps GPU indices for every task.

    Returns ``{"worker": [...], "ps": [...], "colocated": bool, "oversubscribed": bool}`` with ``"cpu"`` /
    ``"cuda:k"`` / ``None`` entries.
    """
    if n_gpus <= 0:
        return {"worker": ["cpu"] * num_workers, "ps": ["cpu"] * num_ps, "colocated": num_ps == 0,
                "oversubscribed": False}
    if shared:
        return {"worker": ["cuda:%d" % (i % n_gpus) for i in range(num_workers)],
                "ps": ["cuda:%d" % ((num_workers + j) % n_gpus) for j in range(num_ps)], "colocated": num_ps == 0,
                "oversubscribed": num_workers + num_ps > n_gpus}
    if num_workers > n_gpus:
        return {"worker": ["cuda:%d" % (i % n_gpus) for i in range(num_workers)], "ps": [None] * num_ps,
                "colocated": True, "oversubscribed": True}
    if num_workers + num_ps <= n_gpus:
        return {"worker": ["cuda:%d" % i for i in range(num_workers)],
                "ps": ["cuda:%d" % (num_workers + j) for j in range(num_ps)], "colocated": num_ps == 0,
                "oversubscribed": False}
    return {"worker": ["cuda:%d" % i for i in range(num_workers)], "ps": [None] * num_ps, "colocated": True,
            "oversubscribed": False}


# --------------------------------------------------------------------------- server
_CURRENT_SERVER = None


def current_server():
    """The task's Server, if the process created one (what strategies attach to)."""
    return _CURRENT_SERVER


class Server:
    """The runtime of one task (``tf.train.Server``).

    Creating it binds the task to its device and, for ps tasks, starts the control-plane listener
    at the task's ``host:port``.  ``join()`` serves parameter-server requests until a shutdown
    request arrives (TF's never returns; the reference's ps process has to be killed,
    /root/reference/example.py:130-131 - here ``request_shutdown`` ends it cleanly).
    """

    def __init__(self, server_or_cluster_def, job_name: Optional[str] = None, task_index: Optional[int] = None,
                 protocol: Optional[str] = None, config: Optional[dict] = None, start: bool = True):
        self.cluster = server_or_cluster_def if isinstance(server_or_cluster_def, ClusterSpec) \
            else ClusterSpec(server_or_cluster_def)
        if job_name is None:
            raise ValueError("job_name must be given")
        self.job_name = job_name
        self.task_index = int(task_index or 0)
        self.address = self.cluster.task_address(job_name, self.task_index)
        self.protocol = protocol or "rccl+tcp"
        self.config = dict(config or {})
        self.control_server = None
        self._started = False
        self._ps_service = None
        global _CURRENT_SERVER
        _CURRENT_SERVER = self
        if start:
            self.start()

    # -- layout helpers
    @property
    def num_ps(self) -> int:
        return self.cluster.num_tasks("ps") if "ps" in self.cluster.jobs else 0

    @property
    def num_workers(self) -> int:
        return self.cluster.num_tasks("worker") if "worker" in self.cluster.jobs else 0

    @property
    def is_chief(self) -> bool:
        return self.job_name == "worker" and self.task_index == 0

    @property
    def world_size(self) -> int:
        return self.num_ps + self.num_workers

    @property
    def rank(self) -> int:
        """Global rank in the data-plane process group: ps tasks first, then workers."""
        return self.task_index if self.job_name == "ps" else self.num_ps + self.task_index

    def worker_rank(self, i: int) -> int:
        return self.num_ps + i

    def ps_rank(self, j: int) -> int:
        return j

    @property
    def target(self) -> str:
        return "dtf://%s" % self.address

    def control_address(self, ps_index: int = 0) -> Tuple[str, int]:
        """Where the control plane of ps task ``ps_index`` listens (worker 0's address if there is no ps)."""
        if self.num_ps > 0:
            return split_host_port(self.cluster.task_address("ps", ps_index))
        return split_host_port(self.cluster.task_address("worker", 0))

    def rendezvous_address(self) -> Tuple[str, int]:
        """TCP address of the ``torch.distributed`` store: host of the first task, a port derived from its
        port (``DTF_RENDEZVOUS_PORT`` overrides)."""
        host, port = self.control_address(0)
        env = os.environ.get("DTF_RENDEZVOUS_PORT")
        rport = int(env) if env else 20000 + (port * 7 + 13) % 20000
        return host, rport

    def _device_plan(self) -> dict:
        import torch
        n = 0
        if not self.config.get("force_cpu") and str(self.config.get("device", "")) != "cpu":
            n = torch.cuda.device_count()           # counting does not initialise the GPU runtime
        shared = os.environ.get("DTF_COMM_BACKEND", "") == "gloo"
        return plan_devices(self.num_ps, self.num_workers, n, shared)

    def gpus_oversubscribed(self) -> bool:
        """True when tasks of this cluster time-share a GPU (more workers than GPUs): RCCL cannot form a group
        then, so collective strategies use gloo and asynchronous owners use hipIpc peer regions."""
        if self.config.get("device"):
            return False
        return bool(self._device_plan()["oversubscribed"])

    def device(self):
        """The device this task computes on (see :func:`plan_devices`).  A ps task without a GPU of its own gets the
        CPU: it then only serves the control plane."""
        import torch
        if self.config.get("device"):
            return torch.device(self.config["device"])
        d = self._device_plan()[self.job_name][self.task_index]
        return torch.device(d if d is not None else "cpu")

    def ps_tasks_have_devices(self) -> bool:
        """False when the ps tasks of this cluster cannot each have a device next to the workers' (more tasks than
        GPUs): the variables are then owned inside the worker processes."""
        if self.num_ps == 0:
            return False
        if self.config.get("device"):
            return True                            # an explicit device per task: the caller has planned the node
        return not self._device_plan()["colocated"]

    # -- lifecycle
    def start(self):
        if self._started:
            return
        serves_control = (self.job_name == "ps") or (self.num_ps == 0 and self.is_chief)
        if serves_control:
            from .control import ControlServer
            host, port = split_host_port(self.address)
            bind_host = "0.0.0.0" if host not in ("localhost", "127.0.0.1") else "127.0.0.1"
            self.control_server = ControlServer(bind_host, port)
        self._started = True

    def join(self):
        """Serve until shutdown.  Only meaningful for ps tasks; a worker that calls it just waits."""
        if self.job_name != "ps":
            import time
            while True:
                time.sleep(3600)
        from .parallel.async_ps import ParameterServerService
        detector = None
        if os.environ.get("DTF_HEARTBEAT", "1") != "0":
            # a parameter server blocked in a transfer with a dead worker would wait forever: the detector ends the
            # process instead, so that the launcher can restart the cluster from the last checkpoint
            from .failure import detector_for_server
            detector = detector_for_server(self, interval_s=float(os.environ.get("DTF_HEARTBEAT_SECS", "0.5")),
                                           max_age_s=float(os.environ.get("DTF_HEARTBEAT_MAX_AGE_SECS", "5")),
                                           abort_after_s=float(os.environ.get("DTF_HEARTBEAT_ABORT_SECS", "10")))
            if detector is not None:
                detector.start()
        self._ps_service = ParameterServerService(self)
        self._ps_service.serve()
        if detector is not None:
            detector.stop(clean=True)

    def shutdown_parameter_servers(self):
        """Ask every ps task to leave ``join()`` (for clusters whose strategy does not use them)."""
        from .control import ControlClient
        for j in range(self.num_ps):
            host, port = self.control_address(j)
            c = ControlClient(host, port, timeout_s=30.0)
            c.push(("ps/req/%d" % j), b'{"op": "shutdown"}')
            c.close()

    def stop(self):
        if self.control_server is not None:
            self.control_server.stop()
            self.control_server = None
        self._started = False

    @staticmethod
    def current() -> Optional["Server"]:
        """The Server this process created (None in single-machine runs)."""
        return _CURRENT_SERVER

    @staticmethod
    def create_local_server(config=None, start=True) -> "Server":
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        return Server({"worker": ["127.0.0.1:%d" % port]}, job_name="worker", task_index=0, config=config,
                      start=start)


def device_and_target(job_name: Optional[str], task_index, ps_hosts: Optional[str], worker_hosts: Optional[str],
                      ps_strategy=None, config: Optional[dict] = None):
    """The reference's ``device_and_target()`` (/root/reference/example.py:108-141) as a library call.

    Returns ``(device_setter, target, server)``; ``(None, "", None)`` for single-machine runs.  A ps
    task never returns from here (it serves until shut down, then the process exits).
    """
    if job_name is None:
        print("Running single-machine training")
        return None, "", None
    print("Running distributed training")
    if task_index is None or task_index == "":
        raise ValueError("Must specify an explicit `task_index`")
    if ps_hosts is None or ps_hosts == "":
        raise ValueError("Must specify an explicit `ps_hosts`")
    if worker_hosts is None or worker_hosts == "":
        raise ValueError("Must specify an explicit `worker_hosts`")
    cluster_spec = ClusterSpec({"ps": ps_hosts.split(","), "worker": worker_hosts.split(",")})
    server = Server(cluster_spec, job_name=job_name, task_index=int(task_index), config=config)
    if job_name == "ps":
        server.join()
        raise SystemExit(0)
    worker_device = "/job:worker/task:{}".format(int(task_index))
    setter = replica_device_setter(worker_device=worker_device, cluster=cluster_spec, ps_strategy=ps_strategy)
    return setter, server.target, server
