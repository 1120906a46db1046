"""Generates tests/golden/rnn_golden.npz by running the REFERENCE's own model classes, imported from
/root/reference (build container only), on seeded inputs with the deterministic synthetic weights of
pepper_thesis_amd.synth.make_weights_p1/p2. Stores inputs and expected outputs (data only; the
weights are regenerated from their seed at test time)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from pepper_thesis_amd import synth  # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)
    blob = {}
    from pepper_variant.modules.python.models.simple_model import TransducerGRU as P1
    for tag, seed, gain in (("p1", 1234, 1.0), ("p1sharp", 77, 3.0)):
        w = synth.make_weights_p1(seed, gain)
        m = P1(26, 1, 256, 28, 3).eval()
        missing = m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
        x = synth.synth_windows(seed + 1, 8)
        taps = {}
        m.encoder.register_forward_hook(lambda mod, i, o: taps.__setitem__("enc", o[0].detach().numpy().copy()))
        m.decoder.register_forward_hook(lambda mod, i, o: taps.__setitem__("dec", o[0].detach().numpy().copy()))
        with torch.no_grad():
            probs = m(torch.from_numpy(x).type(torch.FloatTensor), False).numpy()
        blob[tag + "/seed"] = np.asarray([seed])
        blob[tag + "/gain"] = np.asarray([gain])
        blob[tag + "/images"] = x
        blob[tag + "/probs"] = probs
        blob[tag + "/enc0"] = taps["enc"][0]
        blob[tag + "/dec0"] = taps["dec"][0]
        print(tag, "probs", probs[:3])
    from pepper.modules.python.models.simple_model import TransducerGRU as P2
    for tag, seed, gain in (("p2", 4321, 1.0), ("p2sharp", 99, 4.0)):
        w = synth.make_weights_p2(seed, gain)
        m = P2(1, 10, 1, 128, 5).eval()
        m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
        x = synth.synth_p2_images(seed + 1, 2)
        # the sliding loop of pepper/modules/python/models/predict.py:47-97, driven here
        with torch.no_grad():
            images = torch.from_numpy(x).type(torch.FloatTensor)
            hidden = torch.zeros(images.size(0), 2, 128)
            acc = torch.zeros((images.size(0), images.size(1), 5))
            first_logits = None
            for i in range(0, 1000, 50):
                if i + 100 > 1000:
                    break
                out, hidden = m(images[:, i:i + 100], hidden)
                if first_logits is None:
                    first_logits = out.numpy().copy()
                sm = torch.nn.Sequential(torch.nn.Softmax(dim=2), torch.nn.ZeroPad2d((0, 0, i, 1000 - i - 100)))
                acc = torch.add(acc, sm(out))
            _, labels = torch.max(acc, 2)
        blob[tag + "/seed"] = np.asarray([seed])
        blob[tag + "/gain"] = np.asarray([gain])
        blob[tag + "/images"] = x
        blob[tag + "/labels"] = labels.numpy().astype(np.uint8)
        blob[tag + "/acc"] = acc.numpy()
        blob[tag + "/first_logits"] = first_logits
        blob[tag + "/hidden_final"] = hidden.numpy()
        print(tag, "label histogram", np.bincount(labels.numpy().ravel(), minlength=5))
    path = os.path.join(ROOT, "tests", "golden", "rnn_golden.npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
