"""Does the 5 + 5 topology (XRNERF_TCNN_STRICT_DEFAULTS=1: what tcnn builds if it ignores the config's `num_layers` key) TRAIN like the
(1, 2) one, only later?  Same scene, seeds and schedule; training-batch PSNR (mean of 16 iterations) and rendered PSNR of three frames at
fixed iteration counts, one process per topology.  usage: python tools/topology_psnr.py [iterations]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(iters):
    import torch
    from xrnerf_amd import ops
    from xrnerf_amd.train import Trainer, render_frame, _render_boxes
    dev = torch.device('cuda:0')
    R = 400
    tr = Trainer(dev, n_img=30, H=R, W=R)
    tag = '(%d, %d)' % (tr.net.mlp.density_net.n_hidden, tr.net.mlp.color_net.n_hidden)

    def render_psnr():
        tot = 0.0
        for k in range(3):
            rgb, _ = render_frame(tr.net, tr.data.poses[k], R, R, tr.data.focal)
            o, d = ops.gen_rays(tr.data.poses[k], R, R, tr.data.focal, tr.data.focal, R / 2, R / 2, device=dev)
            gt = _render_boxes(o, d, tr.data.boxes.to(dev))
            tot += float(-10 * torch.log10(((rgb.reshape(-1, 3) - gt[:, :3]) ** 2).mean()))
        return tot / 3
    marks = [m for m in (100, 300, 600, 1000, 2000, 3000, 5000) if m <= iters]
    acc = []
    for it in range(1, iters + 1):
        out = tr.step()
        if any(m - 16 < it <= m for m in marks):
            acc.append(float(out['log_vars']['psnr']))
        if it in marks:
            print('%-22s iter %5d  train PSNR (mean of 16) %6.2f dB  rendered %6.2f dB  rays/batch %6d' %
                  (tag, it, sum(acc) / len(acc), render_psnr(), tr.net.sampler.n_rays_per_batch), flush=True)
            acc = []


if __name__ == '__main__':
    if os.environ.get('TP_CHILD'):
        child(int(os.environ['TP_CHILD']))
    else:
        iters = sys.argv[1] if len(sys.argv) > 1 else '2000'
        for env in (dict(), dict(XRNERF_TCNN_STRICT_DEFAULTS='1')):
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, TP_CHILD=iters, **env), check=False)
