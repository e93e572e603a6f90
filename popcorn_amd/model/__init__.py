from .get_model import Args, calculate_input_channels, get_model_kwargs, model_dict  # noqa: F401
from .popcorn import POPCORN  # noqa: F401
