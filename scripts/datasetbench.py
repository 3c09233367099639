"""Times the device half of the training-input stage (PNG reconstruction + std + augmentation) at B=128."""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import synth                                    # noqa: E402
from birdsoundclassif_amd.nbm_datasets.image_dataset import Img_dataset   # noqa: E402
from birdsoundclassif_amd.nbm_datasets.prepare_dataset import encode_png_gray8   # noqa: E402

B = int(os.environ.get('B', 128))
with tempfile.TemporaryDirectory() as root:
    synth.write_image_dataset(root, encode_png_gray8)
    for host_noise in (True, False):
        ds = Img_dataset(root, transform=True, host_noise=host_noise)
        np.random.seed(0), torch.manual_seed(0)
        t0 = time.perf_counter()
        items = [ds.raw_item(i % 3) for i in range(B)]
        t_host = time.perf_counter() - t0
        ds.collate(items)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            out = ds.collate(items)
        torch.cuda.synchronize()
        t_dev = (time.perf_counter() - t0) / 5
        print(f'host_noise={host_noise}: raw_item {t_host / B * 1e3:.2f} ms/img (inflate + labels + RNG, 1 thread); '
              f'collate B={B}: {t_dev * 1e3:.1f} ms = {B / t_dev:.0f} img/s incl. H2D')
