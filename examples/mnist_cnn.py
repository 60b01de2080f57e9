"""MNIST 2-layer CNN on the parameter server (the "plumbing" configuration: CPU, TCP, -ca y).

    python examples/mnist_cnn.py -m c -cs sync -n 3 -b 64 -ca y --device cpu --transport tcp

``-n`` counts WORKERS; the parameter server is one more process (reference initializer.py:134-145).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from initializer import dataset_fn  # noqa: E402  (the MNIST pipeline of the reference example)
from distributed_tensorflow_amd.cli.launcher import main  # noqa: E402
from distributed_tensorflow_amd.models.mlp import mnist_cnn  # noqa: E402


def model_fn():
    """Conv(32, 3x3, relu) -> MaxPool -> Conv(64, 3x3, relu) -> MaxPool -> Flatten -> Dense(128, relu) -> Dense(10)."""
    return mnist_cnn()


if __name__ == "__main__":
    sys.exit(main(model_fn, dataset_fn))
