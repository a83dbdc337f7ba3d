"""Binding helpers: typed / shaped arrays from graph I/O metadata.

Same names, argument meaning and error behaviour as the reference's `ORT_IO.py` (the L4 layer every
`Inference_*_ONNX.py` imports), restated for the shim's `NodeArg` objects so the reference host loops run
unchanged against this engine. Contract followed (reference file:line):
  numpy_dtype            ORT_IO.py:27-30     "tensor(float)" ... -> numpy dtype (KeyError on unknown)
  is_dynamic_dim         ORT_IO.py:33-34     symbolic (non-int) dimension test
  resolve_shape          ORT_IO.py:37-58     fixed dims win, then `axes[axis]`, then `symbols[name]`
  array_for              ORT_IO.py:61-92     cast + reshape; ValueError when a dynamic axis beyond the value's rank
                                             has no explicit `axes` entry
  filled_for             ORT_IO.py:95-106    np.full of the resolved shape
  scalar_for             ORT_IO.py:109-114   [] or [1] scalar input
  metadata_*             ORT_IO.py:117-141   required-key readers (KeyError / ValueError propagate)
  load_special_token_ids / load_supported_languages / resolve_supported_language   ORT_IO.py:144-181
"""
from __future__ import annotations

import json
from typing import Any, Mapping, Sequence

import numpy as np

_TYPE_TABLE = {
    "tensor(bool)": np.bool_, "tensor(double)": np.float64, "tensor(float)": np.float32, "tensor(float16)": np.float16,
    "tensor(int8)": np.int8, "tensor(int16)": np.int16, "tensor(int32)": np.int32, "tensor(int64)": np.int64,
    "tensor(uint8)": np.uint8, "tensor(uint16)": np.uint16, "tensor(uint32)": np.uint32, "tensor(uint64)": np.uint64,
}


def numpy_dtype(value_or_type: Any) -> np.dtype:
    key = value_or_type if isinstance(value_or_type, str) else value_or_type.type
    return np.dtype(_TYPE_TABLE[key])


def is_dynamic_dim(dim: Any) -> bool:
    return not isinstance(dim, (int, np.integer))


def resolve_shape(value_meta: Any, *, symbols: Mapping[str, int] | None = None,
                  axes: Mapping[int, int] | None = None) -> tuple[int, ...]:
    symbols, axes = symbols or {}, axes or {}
    out = []
    for axis, dim in enumerate(value_meta.shape):
        if not is_dynamic_dim(dim):
            out.append(int(dim))
        elif axes.get(axis) is not None:
            out.append(int(axes[axis]))
        elif isinstance(dim, str) and dim in symbols:
            out.append(int(symbols[dim]))
        else:
            out.append(int(dim))            # raises for an unresolved symbol, like the reference
    return tuple(out)


def array_for(value_meta: Any, value: Any, *, symbols: Mapping[str, int] | None = None,
              axes: Mapping[int, int] | None = None) -> np.ndarray:
    arr = np.asarray(value, dtype=numpy_dtype(value_meta))
    explicit = axes or {}
    dynamic = [axis for axis, dim in enumerate(value_meta.shape) if is_dynamic_dim(dim)]
    unresolved = [axis for axis in dynamic if axis >= arr.ndim and axis not in explicit]
    if unresolved:
        raise ValueError(f"Value for {value_meta.name!r} has rank {arr.ndim}; provide axes for dynamic dimensions "
                         f"{unresolved!r}.")
    runtime = {axis: int(arr.shape[axis]) for axis in dynamic if axis < arr.ndim}
    runtime.update(explicit)
    return np.ascontiguousarray(arr.reshape(resolve_shape(value_meta, symbols=symbols, axes=runtime)))


def filled_for(value_meta: Any, fill_value: Any = 0, *, symbols: Mapping[str, int] | None = None,
               axes: Mapping[int, int] | None = None) -> np.ndarray:
    return np.full(resolve_shape(value_meta, symbols=symbols, axes=axes), fill_value, dtype=numpy_dtype(value_meta))


def scalar_for(value_meta: Any, value: Any) -> np.ndarray:
    if tuple(value_meta.shape) == ():
        return np.asarray(value, dtype=numpy_dtype(value_meta)).reshape(())
    return np.asarray([value], dtype=numpy_dtype(value_meta))


def metadata_by_name(values: Sequence[Any]) -> dict[str, Any]:
    return {v.name: v for v in values}


def metadata_int(metadata: Mapping[str, str], key: str, *, minimum: int | None = None) -> int:
    return int(metadata[key])


def metadata_int_list(metadata: Mapping[str, str], key: str) -> list[int]:
    return [int(tok) for tok in metadata[key].split(",") if tok]


def metadata_json_object(metadata: Mapping[str, str], key: str) -> dict[str, Any]:
    return json.loads(metadata[key])


def load_special_token_ids(metadata: Mapping[str, str]) -> dict[str, Any]:
    return metadata_json_object(metadata, "special_token_ids")


def load_supported_languages(metadata: Mapping[str, str]) -> dict[str, dict[str, Any]]:
    catalog = {}
    for code, raw in metadata_json_object(metadata, "supported_languages").items():
        entry = dict(raw)
        entry["name"] = entry.get("name", code.strip()).strip()
        entry["aliases"] = [a.strip() for a in entry.get("aliases", [])]
        entry["prompt_token_ids"] = entry.get("prompt_token_ids", [])
        catalog[code.strip()] = entry
    return catalog


def resolve_supported_language(catalog: Mapping[str, Mapping[str, Any]], language: str):
    want = language.strip().casefold()
    for code, entry in catalog.items():            # canonical codes take priority over aliases
        if code.casefold() == want:
            return code, entry
    hits = [(code, entry) for code, entry in catalog.items()
            if any(str(a).casefold() == want for a in entry.get("aliases", ()))]
    if len(hits) == 1:
        return hits[0]
    raise ValueError(f"Unsupported language {language!r}; choose one of {sorted(catalog)}.")
