"""Build container only: copies the reference's 5-image Blender-Lego test fixture (real rendered pixels + camera
matrices, /root/reference/test/datasets/data/nerf_synthetic/lego: 2 train, 2 val, 1 test view, 800x800 RGBA) into
oracle/_ref/data/lego.  oracle/_ref/ is git-ignored (nothing of the reference enters the history) but travels to the GPU
box with the gpurun snapshot, like the compiled reference kernels next to it; tools/train_real_lego.py trains on it there."""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = '/root/reference/test/datasets/data/nerf_synthetic/lego'
DST = os.path.join(ROOT, 'oracle', '_ref', 'data', 'lego')

if __name__ == '__main__':
    if not os.path.isdir(SRC):
        sys.exit('no reference tree here')
    for split in ('train', 'val', 'test'):
        os.makedirs(os.path.join(DST, split), exist_ok=True)
        shutil.copy(os.path.join(SRC, 'transforms_%s.json' % split), DST)
        for f in os.listdir(os.path.join(SRC, split)):
            if f.endswith('.png') and 'depth' not in f and 'normal' not in f:
                shutil.copy(os.path.join(SRC, split, f), os.path.join(DST, split, f))
    print(DST, sorted(os.listdir(DST)))
