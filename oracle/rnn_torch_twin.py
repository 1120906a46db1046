"""TEST INFRASTRUCTURE ONLY — a stock-torch.nn assembly of the P1 network for the CPU BASELINE leg of
bench.py. The reference's own CPU inference is exactly this: torch.nn.LSTM / Linear / SELU / Softmax
modules evaluated in eager mode (predict_pytorch, pepper_variant/modules/python/models/
predict_distributed_cpu.py:102-147; model definition models/simple_model.py:23-82); the reference
sources cannot travel to the GPU box, so the same stock modules are assembled here from the state
dict. Never imported by the product path."""
import numpy as np
import torch
import torch.nn as nn


class P1Twin(nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder = nn.LSTM(26, 256, num_layers=1, bidirectional=True, batch_first=True)
        self.decoder = nn.LSTM(512, 256, num_layers=1, bidirectional=True, batch_first=True)
        self.linear_1 = nn.Linear(512 * 33, 512)
        self.linear_2 = nn.Linear(512, 512)
        self.linear_3 = nn.Linear(512, 512)
        self.linear_4 = nn.Linear(512, 512)
        self.linear_5 = nn.Linear(512, 512)
        self.output_layer_type = nn.Linear(512, 3)
        self.activation = nn.SELU()

    def forward(self, x):
        x, _ = self.encoder(x)
        x, _ = self.decoder(x)
        x = torch.flatten(x, start_dim=1, end_dim=2)
        for lin in (self.linear_1, self.linear_2, self.linear_3, self.linear_4, self.linear_5):
            x = self.activation(lin(x))
        return torch.softmax(self.output_layer_type(x), dim=1)


def build_p1(weights: dict) -> P1Twin:
    m = P1Twin().eval()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in weights.items()}, strict=True)
    return m


def predict_p1(model: P1Twin, images: np.ndarray, batch_size: int = 512) -> np.ndarray:
    """the eager predict loop: int8 -> FloatTensor -> model -> numpy (predict_distributed_cpu.py:128-140)"""
    out = []
    with torch.no_grad():
        for i in range(0, images.shape[0], batch_size):
            x = torch.from_numpy(images[i:i + batch_size]).type(torch.FloatTensor)
            out.append(model(x).numpy())
    return np.concatenate(out) if out else np.zeros((0, 3), np.float32)
