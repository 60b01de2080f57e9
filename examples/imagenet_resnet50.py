"""ResNet-50 on synthetic ImageNet-shaped data (224 x 224 x 3, 1000 classes) through the launcher: the headline
configuration as a user script. `bench.py` times the same model and strategy; this file shows the `model_fn` /
`dataset_fn` contract on it.

    python examples/imagenet_resnet50.py -m d -ds keras -n 8 -b 256 --dtype bf16 --max-steps 100
    python examples/imagenet_resnet50.py -m c -cs sync  -n 8 -b 256 --dtype bf16 --max-steps 100

DTF_SYNTH_IMAGENET (default 1024) sets the size of the synthetic image pool each process draws its batches from.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.cli.launcher import main  # noqa: E402
from distributed_tensorflow_amd.data.dataset import Dataset  # noqa: E402
from distributed_tensorflow_amd.models.resnet import resnet50  # noqa: E402


def model_fn():
    return resnet50(1000)


def dataset_fn(batch_size, type='train', shard=True, index=0, buffer_size=10000, reshape=True, n_nodes=None):
    n = int(os.environ.get("DTF_SYNTH_IMAGENET", 1024))
    if type != 'train':
        n = max(batch_size, n // 8)
    rng = np.random.default_rng(1234 if type == 'train' else 4321)
    labels = rng.integers(0, 1000, n).astype(np.int32)
    # class-dependent mean so that the loss can fall: the images are noise around a per-class colour
    tint = np.random.default_rng(99).standard_normal((1000, 1, 1, 3)).astype(np.float32)   # same for train and test
    images = (rng.standard_normal((n, 224, 224, 3), dtype=np.float32) * np.float32(0.5) + tint[labels])   # NHWC
    ds = Dataset.from_tensor_slices((images, labels))
    if shard:
        ds = ds.shard(n_nodes or int(os.environ.get("DTF_N_NODES", "1")), index)
    return ds.batch(batch_size).shuffle(buffer_size)


if __name__ == "__main__":
    sys.exit(main(model_fn, dataset_fn))
