"""Model registry: same surface as the reference's model/get_model.py:19-61 (Args, model_dict,
calculate_input_channels, get_model_kwargs), so ``model_dict[args.model](**get_model_kwargs(args, args.model))``
(run_train.py:60-61, run_eval.py:51-52) constructs the HIP-backed POPCORN."""
from typing import Any, Dict, NamedTuple

from .popcorn import POPCORN


class Args(NamedTuple):
    Sentinel1: bool
    NIR: bool
    Sentinel2: bool
    feature_extractor: str
    occupancymodel: bool
    pretrained: bool
    biasinit: float
    sentinelbuildings: bool


model_dict = {"POPCORN": POPCORN}


def calculate_input_channels(args) -> int:
    """S1 -> +2, NIR -> +1, S2 (RGB) -> +3.  get_model.py:23-32."""
    return 2 * bool(args.Sentinel1) + 1 * bool(args.NIR) + 3 * bool(args.Sentinel2)


_CTOR_FLAGS = ("feature_extractor", "occupancymodel", "pretrained", "biasinit", "sentinelbuildings")


def get_model_kwargs(args, model_name: str) -> Dict[str, Any]:
    """Constructor kwargs of ``model_dict[model_name]`` from the parsed flags (get_model.py:35-61); unknown names raise."""
    if model_name not in model_dict:
        raise ValueError(f"Model {model_name} not found in model dictionary")
    kwargs = {"input_channels": calculate_input_channels(args)}
    kwargs.update((name, getattr(args, name)) for name in _CTOR_FLAGS)
    return kwargs
