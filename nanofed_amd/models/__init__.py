from .gpt2 import GPT2Config, GPT2LM
from .mnist import MNISTMLP, MNISTModel
from .resnet import ResNet18

__all__ = ["MNISTModel", "MNISTMLP", "ResNet18", "GPT2LM", "GPT2Config"]


def build_model(name: str, **kwargs):
    """Model factory used by the engine, the CLI and the benchmark."""
    table = {
        "mnist_cnn": MNISTModel,
        "mnist_mlp": MNISTMLP,
        "resnet18": ResNet18,
        "gpt2_125m": GPT2LM,
    }
    if name not in table:
        raise ValueError(f"Unknown model '{name}'. Known: {sorted(table)}")
    if name == "gpt2_125m":
        return GPT2LM(GPT2Config(**kwargs)) if kwargs else GPT2LM()
    return table[name](**kwargs)
