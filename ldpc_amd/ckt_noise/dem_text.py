"""A reader for stim's detector-error-model text format, so that the decoders that are configured by a ``.dem`` file
(sinter's ``decode_via_files``, the overlapping-window decoders) work in an environment without ``stim``.

Only what ``ckt_noise/dem_matrices.py:122-141`` of the reference consumes is produced: the FLATTENED list of
instructions (``repeat`` blocks unrolled, ``shift_detectors`` applied, so every detector id is absolute), each error as
its probability and its ``^``-separated groups of detector / observable ids, plus ``num_detectors`` and
``num_observables`` (one more than the largest id that occurs, as stim defines them).

Grammar handled (stim ``doc/file_format_dem_detector_error_model.md``)::

    error(p) D0 D3 L1 ^ D4          error[tag](p) ...
    detector(x, y, t) D5            logical_observable L0
    shift_detectors(dx, dy, dt) k   repeat N { ... }        # comments
"""
from __future__ import annotations

import re
from dataclasses import dataclass, field
from typing import List, Tuple

_HEAD = re.compile(r"^([A-Za-z_][A-Za-z_0-9]*)\s*(\[[^\]]*\])?\s*(\(([^)]*)\))?\s*(.*)$")


@dataclass
class FlatError:
    probability: float
    detectors: List[List[int]]    # one list per '^'-separated component, absolute ids
    observables: List[List[int]]


@dataclass
class FlatDem:
    errors: List[FlatError] = field(default_factory=list)
    num_detectors: int = 0
    num_observables: int = 0


def _lines(text: str) -> List[str]:
    out = []
    for raw in text.splitlines():
        line = raw.split("#", 1)[0].strip()
        if not line:
            continue
        # a block may open or close on the line of another instruction: split the braces off
        line = line.replace("{", " {\n").replace("}", "\n}\n")
        out.extend(part.strip() for part in line.split("\n") if part.strip())
    return out


def _parse_block(lines: List[str], pos: int, shift: int, dem: FlatDem) -> Tuple[int, int]:
    """Consume instructions until the closing brace of this block (or the end); returns (position after, detector shift)."""
    while pos < len(lines):
        line = lines[pos]
        if line == "}":
            return pos + 1, shift
        m = _HEAD.match(line)
        if not m:
            raise ValueError(f"cannot parse detector error model line: {line!r}")
        name, args, rest = m.group(1).lower(), m.group(4), m.group(5).strip()
        if name == "repeat":
            if not rest.endswith("{"):
                raise ValueError(f"repeat without a block: {line!r}")
            count = int(rest[:-1].strip())
            body_start = pos + 1
            end = body_start
            for _ in range(count):
                end, shift = _parse_block(lines, body_start, shift, dem)
            if count == 0:  # skip the body
                depth, end = 1, body_start
                while depth:
                    depth += lines[end].endswith("{") - (lines[end] == "}")
                    end += 1
            pos = end
            continue
        targets = rest.split()
        if name == "error":
            if args is None:
                raise ValueError(f"error without a probability: {line!r}")
            dets: List[List[int]] = [[]]
            obs: List[List[int]] = [[]]
            for t in targets:
                if t == "^":
                    dets.append([])
                    obs.append([])
                elif t[0] in "Dd":
                    d = int(t[1:]) + shift
                    dets[-1].append(d)
                    dem.num_detectors = max(dem.num_detectors, d + 1)
                elif t[0] in "Ll":
                    o = int(t[1:])
                    obs[-1].append(o)
                    dem.num_observables = max(dem.num_observables, o + 1)
                else:
                    raise ValueError(f"unknown target {t!r} in {line!r}")
            dem.errors.append(FlatError(float(args.split(",")[0]), dets, obs))
        elif name == "detector":
            for t in targets:
                dem.num_detectors = max(dem.num_detectors, int(t[1:]) + shift + 1)
        elif name == "logical_observable":
            for t in targets:
                dem.num_observables = max(dem.num_observables, int(t[1:]) + 1)
        elif name == "shift_detectors":
            shift += int(targets[0]) if targets else 0
        else:
            raise NotImplementedError(f"detector error model instruction {name!r}")
        pos += 1
    return pos, shift


def parse_dem_text(text: str) -> FlatDem:
    dem = FlatDem()
    _parse_block(_lines(text), 0, 0, dem)
    return dem


def load_dem(source) -> FlatDem:
    """``source``: DEM text, a path to a ``.dem`` file, a ``FlatDem``, or a ``stim.DetectorErrorModel`` (its ``str`` is the text)."""
    import os
    import pathlib
    if isinstance(source, FlatDem):
        return source
    if isinstance(source, pathlib.Path) or (isinstance(source, str) and "\n" not in source and os.path.exists(source)):
        with open(source, "r") as f:
            return parse_dem_text(f.read())
    if isinstance(source, str):
        return parse_dem_text(source)
    return parse_dem_text(str(source))  # stim.DetectorErrorModel prints as its file format
