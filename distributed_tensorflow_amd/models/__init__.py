"""Model API: the Keras-style ``Sequential`` and ready-made model builders."""
from .sequential import LocalStrategy, Sequential, default_device, load_model, model_from_json  # noqa: F401
from .zoo import mnist_cnn, mnist_mlp, wide_mlp, wide_mlp_learning_rate, xor_mlp  # noqa: F401
