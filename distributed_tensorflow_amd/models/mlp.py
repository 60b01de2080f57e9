"""The reference's example models, written against this framework's Keras-style layers.

``reference_mlp`` is the launcher's example ``model_fn``: ``Flatten(28, 28) ->
Dense(512, relu) -> Dropout(0.2) -> Dense(10, softmax)``, 407,050 parameters
(reference initializer.py:12-21). ``mnist_cnn`` is the 2-layer CNN used by the
"MNIST CNN, centralized sync PS, CPU" configuration.
"""
from __future__ import annotations

from ..api.layers import Conv2D, Dense, Dropout, Flatten, MaxPooling2D
from ..api.model import Sequential


def reference_mlp(input_shape=(28, 28), hidden: int = 512, classes: int = 10, dropout: float = 0.2) -> Sequential:
    return Sequential([
        Flatten(input_shape=input_shape),
        Dense(hidden, activation="relu"),
        Dropout(dropout),
        Dense(classes, activation="softmax"),
    ])


def mnist_cnn(input_shape=(28, 28, 1), classes: int = 10) -> Sequential:
    """Two 3x3 convolution layers with max pooling, then a dense classifier (~1.2 M parameters)."""
    return Sequential([
        Conv2D(32, 3, activation="relu", input_shape=input_shape),
        MaxPooling2D(2),
        Conv2D(64, 3, activation="relu"),
        MaxPooling2D(2),
        Flatten(),
        Dense(128, activation="relu"),
        Dropout(0.2),
        Dense(classes, activation="softmax"),
    ])
