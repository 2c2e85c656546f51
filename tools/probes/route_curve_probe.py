import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch.utils.data import DataLoader
from nanofed_amd.data import synthetic_mnist
from nanofed_amd.data.mnist import MNISTTensorDataset
from nanofed_amd.models import MNISTModel
from nanofed_amd.trainer import TorchTrainer, TrainingConfig as TC

def run(native_on, seed):
    os.environ["NANOFED_NATIVE_TRAINER"] = "1" if native_on else "0"
    images, labels = synthetic_mnist(2048, seed=5)
    torch.manual_seed(0)
    loader = DataLoader(MNISTTensorDataset(images, labels), batch_size=64, shuffle=True)
    trainer = TorchTrainer(TC(epochs=8, batch_size=64, learning_rate=0.05, device="cuda"))
    torch.manual_seed(seed)
    model = MNISTModel().to("cuda")
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    return [round(trainer.train_epoch(model, loader, opt, e).loss, 3) for e in range(8)]

for seed in (1, 2, 3):
    print(seed, "native", run(True, seed), flush=True)
    print(seed, "torch ", run(False, seed), flush=True)
