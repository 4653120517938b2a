"""Detector error model -> check matrix, observables matrix, priors (reference: ckt_noise/dem_matrices.py:52-171).

Same outputs as the reference's ``detector_error_model_to_check_matrices`` -- columns are the distinct detector sets in
order of first appearance (:87-92), probabilities of repeated sets are combined as independent flips (:93), the
observables of a set are those of its LAST occurrence (:92), edges are the ``^``-components with at most two detectors
(:96-113) -- but the model may be given as text or a file path as well as a ``stim.DetectorErrorModel``: the text is
read by ``ldpc_amd.ckt_noise.dem_text`` and ``stim`` is not needed.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
from scipy.sparse import csc_matrix

from ldpc_amd.ckt_noise.dem_text import load_dem


@dataclass
class DemMatrices:
    check_matrix: csc_matrix
    observables_matrix: csc_matrix
    edge_check_matrix: csc_matrix
    edge_observables_matrix: csc_matrix
    hyperedge_to_edge_matrix: csc_matrix
    priors: np.ndarray


def _odd_members(groups) -> frozenset:
    """Ids that occur an odd number of times over the groups (each group first reduced to a set, as the reference does)."""
    acc = set()
    for g in groups:
        acc ^= set(g)
    return frozenset(acc)


def _columns_to_csc(columns, shape) -> csc_matrix:
    """``columns[j]`` = row ids of column j."""
    rows = [r for j in range(len(columns)) for r in columns[j]]
    cols = [j for j in range(len(columns)) for _ in columns[j]]
    return csc_matrix((np.ones(len(rows), dtype=np.uint8), (np.asarray(rows, dtype=np.int64), np.asarray(cols, dtype=np.int64))),
                      shape=shape)


def detector_error_model_to_check_matrices(dem, allow_undecomposed_hyperedges: bool = False) -> DemMatrices:
    flat = load_dem(dem)
    column_of = {}          # detector set -> hyperedge id
    column_dets, column_obs, priors, column_edges = [], [], [], []
    edge_of = {}            # detector set (<= 2 members) -> edge id
    edge_dets, edge_obs = [], []
    for err in flat.errors:
        dets = _odd_members(err.detectors)
        hid = column_of.get(dets)
        if hid is None:
            hid = column_of[dets] = len(column_dets)
            column_dets.append(dets)
            column_obs.append(frozenset())
            priors.append(0.0)
            column_edges.append(None)
        column_obs[hid] = _odd_members(err.observables)
        q = priors[hid]
        priors[hid] = q * (1 - err.probability) + err.probability * (1 - q)
        eids = []
        for part_d, part_o in zip(err.detectors, err.observables):
            part = frozenset(part_d)
            if len(part) > 2:
                if not allow_undecomposed_hyperedges:
                    raise ValueError(
                        "A hyperedge error mechanism was found that was not decomposed into edges. "
                        "This can happen if you do not set `decompose_errors=True` as required when "
                        "calling `circuit.detector_error_model`.")
                continue
            eid = edge_of.get(part)
            if eid is None:
                eid = edge_of[part] = len(edge_dets)
                edge_dets.append(part)
                edge_obs.append(frozenset())
            edge_obs[eid] = frozenset(part_o)
            eids.append(eid)
        if column_edges[hid] is None:
            column_edges[hid] = frozenset(eids)
    nh, ne = len(column_dets), len(edge_dets)
    return DemMatrices(
        check_matrix=_columns_to_csc(column_dets, (flat.num_detectors, nh)),
        observables_matrix=_columns_to_csc(column_obs, (flat.num_observables, nh)),
        edge_check_matrix=_columns_to_csc(edge_dets, (flat.num_detectors, ne)),
        edge_observables_matrix=_columns_to_csc(edge_obs, (flat.num_observables, ne)),
        hyperedge_to_edge_matrix=_columns_to_csc([c if c is not None else frozenset() for c in column_edges], (ne, nh)),
        priors=np.asarray(priors, dtype=np.float64),
    )
