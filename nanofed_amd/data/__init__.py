from .cifar import (
    CIFAR10_MEAN,
    CIFAR10_STD,
    load_cifar10_arrays,
    load_cifar10_data,
    write_cifar10_batch,
)
from .mnist import load_mnist_arrays, load_mnist_data, normalize_mnist, read_idx
from .partition import (
    dirichlet_partition,
    iid_partition,
    label_histogram,
    sized_partition,
)
from .tokens import (
    find_token_file,
    load_token_sequences,
    open_token_stream,
    token_windows,
    write_token_file,
)
from .synthetic import (
    synthetic_cifar10,
    synthetic_cifar10_hard,
    synthetic_images,
    synthetic_mnist,
    synthetic_mnist_hard,
    synthetic_tokens,
)

__all__ = [
    "load_mnist_data",
    "load_mnist_arrays",
    "normalize_mnist",
    "read_idx",
    "iid_partition",
    "sized_partition",
    "dirichlet_partition",
    "label_histogram",
    "synthetic_images",
    "synthetic_mnist",
    "synthetic_mnist_hard",
    "synthetic_cifar10",
    "synthetic_cifar10_hard",
    "load_cifar10_arrays",
    "load_cifar10_data",
    "write_cifar10_batch",
    "CIFAR10_MEAN",
    "CIFAR10_STD",
    "synthetic_tokens",
    "find_token_file",
    "load_token_sequences",
    "open_token_stream",
    "token_windows",
    "write_token_file",
]
