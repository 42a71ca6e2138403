"""Golden vector for the tail of the input pipeline, produced with the very operations the
reference's loader applies (PIL crop + FLIP_LEFT_RIGHT from transforms/joint_transforms.py,
MaskToTensor from transforms/transforms.py, and torchvision's ToTensor/Normalize restated with
torch -- torchvision is not in this image): a seeded 40x56 RGB image and label map, two windows.
    python tests/golden/make_golden_pipeline.py   ->   pipeline_golden.pt"""
import os

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]          # config.py:96-97


def main():
    rng = np.random.RandomState(5)
    img = rng.randint(0, 256, (40, 56, 3)).astype(np.uint8)
    lab = rng.randint(0, 19, (40, 56)).astype(np.uint8)
    lab[rng.rand(40, 56) < 0.1] = 255
    cases = []
    for (x0, y0, w, h), flip in (((3, 5, 32, 24), False), ((0, 0, 56, 40), True), ((17, 9, 16, 8), True)):
        pi, pm = Image.fromarray(img), Image.fromarray(lab)
        pi, pm = pi.crop((x0, y0, x0 + w, y0 + h)), pm.crop((x0, y0, x0 + w, y0 + h))
        if flip:
            pi, pm = pi.transpose(Image.FLIP_LEFT_RIGHT), pm.transpose(Image.FLIP_LEFT_RIGHT)
        t = torch.from_numpy(np.array(pi)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)   # ToTensor
        t = t.sub(torch.tensor(MEAN).view(3, 1, 1)).div(torch.tensor(STD).view(3, 1, 1))             # Normalize
        m = torch.from_numpy(np.array(pm, dtype=np.int32)).long()                                    # MaskToTensor
        cases.append({"window": (x0, y0, w, h), "flip": flip, "image": t, "labels": m})
    torch.save({"img": torch.from_numpy(img), "lab": torch.from_numpy(lab), "mean": MEAN, "std": STD, "cases": cases},
               os.path.join(HERE, "pipeline_golden.pt"))
    print("cases", len(cases))


if __name__ == "__main__":
    main()
