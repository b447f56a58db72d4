"""JSON (de)serialisation of ABI batches and results for the committed fixtures under tests/golden/."""
from __future__ import annotations

import json

import numpy as np

from evergreen_amd import abi


def _rows(a):
    return {n: a[n].tolist() for n in a.dtype.names}


def _unrows(d, dtype):
    n = len(next(iter(d.values()))) if d else 0
    out = np.zeros(n, dtype)
    for k, v in d.items():
        out[k] = v
    return out


def batch_to_json(b: abi.PlanBatch) -> dict:
    j = {"n_distros": b.n_distros, "now_ns": b.now_ns, "cols": {k: v.tolist() for k, v in b.cols.items()},
         "dep_off": b.dep_off.tolist(), "edges": {k: v.tolist() for k, v in b.edges.items()}, "distros": _rows(b.distros),
         "task_off": b.task_off.tolist(), "tg_off": b.tg_off.tolist(), "ver_off": b.ver_off.tolist()}
    if b.alloc_params is not None:
        j["alloc_params"] = _rows(b.alloc_params)
        j["host_off"] = b.host_off.tolist()
        j["hosts"] = {k: v.tolist() for k, v in b.hosts.items()}
    if b.tg_name_key is not None:
        j["tg_name_key"] = b.tg_name_key.tolist()
    return j


def batch_from_json(j: dict) -> abi.PlanBatch:
    b = abi.PlanBatch(
        n_distros=j["n_distros"], now_ns=j["now_ns"],
        cols={k: np.asarray(j["cols"][k], dt) for k, dt in abi.TASK_COLUMNS.items()},
        dep_off=np.asarray(j["dep_off"], np.int32),
        edges={k: np.asarray(j["edges"][k], dt) for k, dt in abi.EDGE_COLUMNS.items()},
        distros=_unrows(j["distros"], abi.DISTRO_PARAMS_DTYPE), task_off=np.asarray(j["task_off"], np.int32),
        tg_off=np.asarray(j["tg_off"], np.int32), ver_off=np.asarray(j["ver_off"], np.int32),
        tg_name_key=np.asarray(j["tg_name_key"], np.int32) if "tg_name_key" in j else None)
    if "alloc_params" in j:
        b.alloc_params = _unrows(j["alloc_params"], abi.ALLOC_PARAMS_DTYPE)
        b.host_off = np.asarray(j["host_off"], np.int32)
        b.hosts = {k: np.asarray(j["hosts"][k], dt) for k, dt in abi.HOST_COLUMNS.items()}
    b.check()
    return b


def plan_to_json(r: abi.PlanResult) -> dict:
    return {"order": r.order.tolist(), "breakdown": r.breakdown.tolist(), "deps_met": r.deps_met.tolist(),
            "wait_ns": r.wait_ns.tolist(), "distro_info": _rows(r.distro_info), "group_info": _rows(r.group_info),
            "n_units": r.n_units.tolist()}


def plan_from_json(j: dict) -> abi.PlanResult:
    return abi.PlanResult(order=np.asarray(j["order"], np.int32), breakdown=np.asarray(j["breakdown"], np.int64).reshape(-1, abi.BREAKDOWN_FIELDS),
                          deps_met=np.asarray(j["deps_met"], np.uint8), wait_ns=np.asarray(j["wait_ns"], np.int64),
                          distro_info=_unrows(j["distro_info"], abi.DISTRO_INFO_DTYPE),
                          group_info=_unrows(j["group_info"], abi.GROUP_INFO_DTYPE), n_units=np.asarray(j["n_units"], np.int32))


def load(path):
    with open(path) as f:
        return json.load(f)


def dump(obj, path):
    with open(path, "w") as f:
        json.dump(obj, f, separators=(",", ":"))
        f.write("\n")
