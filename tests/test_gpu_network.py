"""End-to-end checks behind the registry on the GPU: the fused single-node train step equals the modular
autograd path (same kernels), training reduces the loss, rendering works, state_dict round-trips."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def make(dev, seed=0):
    from xrnerf_amd.train import Trainer
    return Trainer(dev, n_img=2, H=96, W=96, seed=seed, ema=False)


def test_fused_step_equals_modular_step(dev, monkeypatch):
    a, b = make(dev), make(dev)
    b.net.load_state_dict(a.net.state_dict())
    out = {}
    for name, tr in (('fused', a), ('modular', b)):
        monkeypatch.setenv('XRNERF_STEP', name)
        tr.net.sampler.set_iter(0)
        batch = {k: v[None] for k, v in tr.data.next_batch().items()}
        o = tr.net.train_step(batch, tr.opt)
        o['loss'].backward()
        out[name] = (float(o['loss']), float(o['log_vars']['psnr']),
                     [p.grad.clone() for p in tr.net.parameters() if p.grad is not None])
    lf, pf, gf = out['fused']; lm, pm, gm = out['modular']
    assert abs(lf - lm) <= 1e-5 * abs(lm) and abs(pf - pm) <= 1e-4
    assert len(gf) == len(gm) == 3
    for x, y in zip(gf, gm):
        assert float((x - y).abs().max()) <= 1e-5 * max(1.0, float(y.abs().max()))


def test_bucketed_gradient_sync_path_covers_every_gradient(dev):
    """The data-parallel fused step hands its gradients to BucketedGradSync in three buckets while it is
    still producing them; with a recording stand-in (no peers) the buckets must tile the parameter set and
    the resulting gradients must equal the single-GPU step's (scaled by the factor finish() returns)."""
    from xrnerf_amd import dist as xd

    class Recorder(xd.BucketedGradSync):
        def __init__(self):
            super().__init__(1)
            self.sizes = []

        def ready(self, bucket):
            assert bucket.is_contiguous()
            self.sizes.append(bucket.numel())

        def finish(self):
            return 0.5

    a, b = make(dev), make(dev)
    b.net.load_state_dict(a.net.state_dict())
    b.net.grad_sync = Recorder()
    grads = []
    for tr in (a, b):
        tr.net.sampler.set_iter(0)
        batch = {k: v[None] for k, v in tr.data.next_batch().items()}
        tr.net.train_step(batch, tr.opt)['loss'].backward()
        grads.append([p.grad.clone() for p in tr.net.parameters() if p.grad is not None])
    n_mlp = a.net.mlp.density_net.params.numel() + a.net.mlp.color_net.params.numel()
    n_tab = a.net.mlp.embedder_pos.params.numel()
    sizes = b.net.grad_sync.sizes
    assert sizes[0] == n_mlp and sum(sizes[1:]) == n_tab and len(sizes) == 3 and sizes[1] == 8 * 2 * (1 << 19)
    for x, y in zip(grads[0], grads[1]):
        assert float((0.5 * x - y).abs().max()) <= 1e-5 * max(1.0, float(x.abs().max()))


def test_training_reduces_loss_and_renders(dev):
    from xrnerf_amd.train import render_frame
    tr = make(dev)
    first = float(tr.step()['log_vars']['loss'])
    for _ in range(60):
        out = tr.step()
    last = float(out['log_vars']['loss'])
    assert np.isfinite(last) and last < 0.7 * first
    assert tr.net.sampler.n_rays_per_batch % 128 == 0
    rgb, alpha = render_frame(tr.net, tr.data.poses[0], 96, 96, tr.data.focal)
    assert rgb.shape == (96, 96, 3) and alpha.shape == (96, 96, 1)
    assert torch.isfinite(rgb).all() and 0.0 <= float(alpha.min()) and float(alpha.max()) <= 1.0 + 1e-5
    # config chunking (networks/nerf.py:50-69) gives the same image as one big chunk
    rgb2, _ = render_frame(tr.net, tr.data.poses[0], 96, 96, tr.data.focal, chunk=4096)
    # the per-launch jitter stream advances with every K1 call (hidden RNG of the reference): compare loosely
    assert float((rgb - rgb2).abs().mean()) < 0.05
    sd = tr.net.state_dict()
    tr2 = make(dev, seed=5)
    tr2.net.load_state_dict(sd)
    assert torch.equal(tr2.net.mlp.embedder_pos.params, tr.net.mlp.embedder_pos.params)


def test_window_march_matches_marches_in_place(dev):
    """The marches of a refresh window issued right behind the refresh as one series of launches on a side stream
    (Trainer.march_window = 'side') must not change what is computed: same sample counts exactly, same losses up to
    summation-order rounding."""
    a, b = make(dev), make(dev)
    b.net.load_state_dict(a.net.state_dict())
    a.march_window, b.march_window = 'side', 'off'
    la, lb, na, nb = [], [], [], []
    for _ in range(40):
        # (read at once: the logged loss is a view of the step's recycled buffers)
        la.append(float(a.step()['log_vars']['loss'])); na.append(int(a.net.sampler.n_valid_dev))
        lb.append(float(b.step()['log_vars']['loss'])); nb.append(int(b.net.sampler.n_valid_dev))
    torch.cuda.synchronize()
    # the same batches, the same samples, the same (atomic-free, fixed-order) arithmetic: the whole trajectory agrees
    assert na == nb
    la, lb = np.array(la), np.array(lb)
    assert np.abs(la - lb).max() <= 1e-5 * np.abs(lb).max()
    assert a.net.sampler.n_rays_per_batch == b.net.sampler.n_rays_per_batch


def test_early_terminated_render_within_eps_of_full_render(dev):
    """optional ERT path: identical to the reference-behaviour render to within eps = 1e-4 (north_star's RGB bar)"""
    from xrnerf_amd.train import render_frame, render_frame_ert
    tr = make(dev)
    for _ in range(150):
        tr.step()
    sampler = tr.net.sampler
    sampler.rewind_marches()                     # (what the first test-mode launch would do: the window's marches ahead are taken back)
    calls = sampler.k1_calls
    full_rgb, full_a = render_frame(tr.net, tr.data.poses[1], 96, 96, tr.data.focal)
    sampler.k1_calls = calls                     # same hidden-RNG position -> same jittered samples
    ert_rgb, ert_a = render_frame_ert(tr.net, tr.data.poses[1], 96, 96, tr.data.focal, eps=1e-4)
    assert float((full_rgb - ert_rgb).abs().max()) <= 1.5e-4 and float((full_a - ert_a).abs().max()) <= 1.5e-4
    ev, total = render_frame_ert.last_evaluated
    assert 0 < ev <= total
    sampler.k1_calls = calls                     # eps = 0 never terminates: bit-for-bit the same integration order
    z_rgb, z_a = render_frame_ert(tr.net, tr.data.poses[1], 96, 96, tr.data.focal, eps=-1.0)
    assert float((full_rgb - z_rgb).abs().max()) <= 2e-6 and render_frame_ert.last_evaluated[0] == total
    # the same path behind the registry's frame entry point (XRNERF_FRAME=ert: HashNerfNetwork.batchify_forward, i.e. val_step / test_step)
    import os
    from xrnerf_amd import ops
    o, d = ops.gen_rays(tr.data.poses[1], 96, 96, tr.data.focal, tr.data.focal, 48.0, 48.0, device=dev)
    os.environ['XRNERF_FRAME'] = 'ert'
    try:
        sampler.k1_calls = calls
        with torch.no_grad():
            ret = tr.net.batchify_forward({'rays_o': o, 'rays_d': d, 'img_ids': torch.zeros((o.shape[0], 1), dtype=torch.int32, device=dev)}, is_test=True)
    finally:
        os.environ.pop('XRNERF_FRAME', None)
    assert torch.equal(ret['rgb'].reshape(96, 96, 3), ert_rgb) and torch.equal(ret['alpha'].reshape(96, 96, 1), ert_a)


def test_training_converges_on_the_synthetic_scene(dev):
    """end to end behind the registry: a few hundred iterations on the analytic Lego-shaped scene reach a
    rendered-frame PSNR well above the untrained level (~12 dB) -- gradients, optimiser and grid upkeep cooperate"""
    from xrnerf_amd import ops
    from xrnerf_amd.train import Trainer, render_frame, _render_boxes
    R = 160
    tr = Trainer(dev, n_img=12, H=R, W=R, seed=0)
    for _ in range(500):
        tr.step()
    rgb, _ = render_frame(tr.net, tr.data.poses[0], R, R, tr.data.focal)
    o, d = ops.gen_rays(tr.data.poses[0], R, R, tr.data.focal, tr.data.focal, R / 2, R / 2, device=dev)
    gt = _render_boxes(o, d, tr.data.boxes.to(dev))[:, :3]
    psnr = float(-10 * torch.log10(((rgb.reshape(-1, 3) - gt) ** 2).mean()))
    assert psnr > 24.0, psnr


def test_blender_scene_directory_trains_through_the_device_ray_table(dev, O, tmp_path):
    """datasets.HashNerfDataset: a Blender-format directory -> device-resident ray table in
    the reference's layout and image order (val, train) -> sampler hand-off -> training steps."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_datasets import write_scene
    from xrnerf_amd import synthetic as S
    from xrnerf_amd.datasets import HashNerfDataset, load_blender_data
    from xrnerf_amd.train import Trainer
    write_scene(str(tmp_path), H=16, W=12, n=(3, 2, 2))
    cfg = dict(datadir=str(tmp_path), half_res=False, testskip=1, white_bkgd=False, load_alpha=True,
               N_rand_per_sampler=256, mode='val', val_n=1)
    ds = HashNerfDataset(cfg, device=dev)                       # 'val' = unshuffled table
    imgs, poses, _, hwf, i_split = load_blender_data(str(tmp_path))
    order = np.concatenate((i_split[1], i_split[0]))            # (val, train), hashnerf_dataset.py:33
    assert ds.n_img == 5 and ds.rays_rgb.shape == (5 * 16 * 12, 11)
    table = ds.rays_rgb.cpu().numpy().reshape(5, 16 * 12, 11)
    pose_ngp = S.poses_nerf2ngp(poses[order])
    for k in (0, 3):
        o, d = O.gen_rays(pose_ngp[k], 16, 12, hwf[2], hwf[2], 6.0, 8.0)
        assert np.abs(table[k, :, :3] - o).max() <= 1e-6 and np.abs(table[k, :, 3:6] - d).max() <= 2.4e-7
        assert np.array_equal(table[k, :, 6:10], imgs[order[k]].reshape(-1, 4))
        assert np.all(table[k, :, 10] == k)
    val = ds.fetch_val_data()
    assert val['poses'].shape == (1, 4, 3) and val['images'].shape == (1, 16, 12, 4)
    # train mode: shuffled table, same multiset of rows; steps run and the hand-off reaches the sampler
    tds = HashNerfDataset(dict(cfg, mode='train'), device=dev)
    assert torch.equal(tds.rays_rgb.sum(0), ds.rays_rgb.sum(0)) or torch.allclose(tds.rays_rgb.sum(0), ds.rays_rgb.sum(0), rtol=1e-4)
    tr = Trainer(dev, dataset=tds, ema=False)
    assert tr.net.sampler.n_rays_per_batch == 4096 or tr.net.sampler.n_rays_per_batch > 0
    for _ in range(3):
        out = tr.step()
    assert np.isfinite(float(out['log_vars']['loss']))


def test_gradient_buffers_survive_callers_that_keep_their_gradients(dev):
    """The native fused step WRITES its gradients into two recycled buffer sets that become `.grad`.  A caller that keeps `.grad`
    alive -- `zero_grad(set_to_none=False)`, or accumulation over several backward passes -- must still see torch semantics:
    three accumulated steps equal the sum of the three steps' gradients taken one by one (the launch sequence XRNERF_STEP=py
    allocates fresh gradient tensors every step and is the reference here)."""
    import os
    from xrnerf_amd.train import Trainer

    def grads_of(tr, n_steps, accumulate, keep):
        net, out = tr.net, []
        params = [p for p in net.parameters() if p.numel() > 0]
        for p in params:
            p.grad = None
        for _ in range(n_steps):
            net.sampler.set_iter(3)                                  # no grid refresh, no batch-size change
            b = tr.data.next_batch()
            o = net.train_step({k: v[None] for k, v in b.items()}, None)
            if not accumulate:
                if keep:
                    for p in params:
                        if p.grad is not None:
                            p.grad.zero_()                           # optimizer.zero_grad(set_to_none=False)
                else:
                    for p in params:
                        p.grad = None
            o['loss'].backward()
            out.append([p.grad.detach().clone() for p in params])
        return out

    def fresh(py):
        if py:
            os.environ['XRNERF_STEP'] = 'py'
        else:
            os.environ.pop('XRNERF_STEP', None)
        tr = Trainer(dev, n_img=3, H=128, W=128, ema=False)
        tr.march_window = 'off'
        tr.net.sampler.on_sampled = None
        for _ in range(2):
            tr.step()                                                # past the first refresh: a real occupancy grid
        tr.data.cur_i = 0
        tr.data.batches_drawn = 0
        tr.net.sampler.k1_calls = 100
        return tr
    try:
        ref = grads_of(fresh(True), 3, accumulate=False, keep=False)
        want_acc = [sum(g[i] for g in ref) for i in range(len(ref[0]))]
        kept = grads_of(fresh(False), 3, accumulate=False, keep=True)    # zero_grad(set_to_none=False) between the steps
        acc = grads_of(fresh(False), 3, accumulate=True, keep=True)
    finally:
        os.environ.pop('XRNERF_STEP', None)
    for i in range(len(want_acc)):
        scale = float(want_acc[i].abs().max())
        for s in range(3):
            assert float((kept[s][i] - ref[s][i]).abs().max()) <= 1e-4 * scale, (i, s)
        assert float((acc[2][i] - want_acc[i]).abs().max()) <= 1e-4 * scale, i


@pytest.mark.parametrize('trained_steps', [0, 40])
def test_chunked_frame_without_per_chunk_readback_gives_the_same_pixels(dev, trained_steps, monkeypatch):
    """The registry's frame path (val_step -> batchify_forward in chunk = 4096 pieces, networks/nerf.py:50-69) marched without a
    host read-back per chunk: identical pixels to the synchronous form.  trained_steps = 0: the dense initial occupancy grid
    gives > 48 samples per ray, so chunks OVERFLOW their estimated buffers and are done again (same RNG call index)."""
    from xrnerf_amd.train import Trainer, render_frame
    tr = Trainer(dev, n_img=3, H=128, W=128, ema=False)
    for _ in range(max(trained_steps, 1)):
        tr.step()
    net, pose = tr.net, tr.data.poses[1]
    H = W = 160                                                    # 25 600 rays = 7 chunks of 4096
    net.sampler.rewind_marches()                                   # (a test-mode launch takes the marches issued ahead back first)
    k1 = net.sampler.k1_calls
    monkeypatch.setenv('XRNERF_FRAME', 'sync')
    rgb_s, a_s = render_frame(net, pose, H, W, tr.data.focal * H / 128, chunk=4096)             # the reference's loop, one read-back per chunk
    calls = net.sampler.k1_calls - k1
    for attempt in range(2):                                       # second attempt: buffers sized from the first frame's rows per ray
        monkeypatch.setenv('XRNERF_FRAME', 'async')
        net.sampler.k1_calls = k1
        rgb_a, a_a = render_frame(net, pose, H, W, tr.data.focal * H / 128, chunk=4096)
        assert net.sampler.k1_calls - k1 == calls == 7
        assert torch.equal(rgb_a, rgb_s) and torch.equal(a_a, a_s), attempt
    # the default: the whole frame as one launch per kernel, K1 drawing each ray's jitter as its chunk's launch would
    monkeypatch.delenv('XRNERF_FRAME')
    net.sampler.k1_calls = k1
    rgb_1, a_1 = render_frame(net, pose, H, W, tr.data.focal * H / 128, chunk=4096)
    assert net.sampler.k1_calls - k1 == 7
    assert torch.equal(rgb_1, rgb_s) and torch.equal(a_1, a_s)
    assert float(a_s.max()) > 0.5
    # a chunk whose rays all miss the occupied cells (sky rows): zero samples, background pixels, both forms
    up = np.array(pose, dtype=np.float32).copy()
    up[3] = [0.5, -5.0, 0.5]                                       # camera far outside, looking away: no ray enters the cube
    for env in ('sync', 'async'):
        monkeypatch.setenv('XRNERF_FRAME', env)
        rgb_e, a_e = render_frame(net, up, 96, 96, tr.data.focal, chunk=4096)
        assert float(a_e.abs().max()) == 0.0


def test_table_update_inside_the_scatter_equals_scatter_plus_optimiser_launch(dev, monkeypatch):
    """One GPU: the trainer lets the table scatter apply FusedAdam's update to the hash table (xr_hashgrid_bwd_adam) instead of
    writing a gradient the optimiser launch reads back.  20 iterations (two grid refreshes) with and without: parameters, Adam
    moments and EMA copies bit for bit, and no table gradient is produced on the fused path."""
    from xrnerf_amd.train import Trainer
    out = []
    for fuse in ('1', '0'):
        tr = Trainer(dev, n_img=3, H=128, W=128, seed=3, fuse_adam=fuse == '1')
        assert tr.fuse_adam == (fuse == '1')
        for _ in range(20):
            tr.step()
        torch.cuda.synchronize()
        table = tr.net.mlp.embedder_pos.params
        st = tr.opt.state[table]
        assert (table.grad is None) == (fuse == '1')
        out.append([p.detach().clone() for p in tr.net.parameters()] + [st['m'].clone(), st['v'].clone(), st['ema'].clone(), st['step']])
    for a, b in zip(out[0][:-1], out[1][:-1]):
        assert torch.equal(a, b)
    assert out[0][-1] == out[1][-1] == 20


def test_refresh_samples_generated_one_iteration_early_leave_the_trajectory_alone(dev, monkeypatch):
    """prefetch_k6 (default on): K6 and the clear of the temporary grid of a refresh run on the side stream during the
    iteration before it.  40 iterations (refreshes at 0, 16, 32) with and without: bit-identical parameters, grids, RNG counters."""
    from xrnerf_amd.train import Trainer
    out = []
    for on in ('1', '0'):
        tr = Trainer(dev, n_img=3, H=128, W=128, seed=5, prefetch_k6=on == '1')
        assert tr.prefetch_k6 == (on == '1')
        for _ in range(40):
            tr.step()
        torch.cuda.synchronize()
        out.append(([p.detach().clone() for p in tr.net.parameters()], tr.net.sampler.density_grid.clone(),
                    tr.net.sampler.density_grid_bitfield.clone(), tr.net.sampler.k6_calls, tr.net.sampler.n_rays_per_batch))
    (pa, ga, ba, ka, na), (pb, gb, bb, kb, nb) = out
    assert ka == kb and na == nb
    for a, b in zip(pa, pb):
        assert torch.equal(a, b)
    assert torch.equal(ga, gb) and torch.equal(ba, bb)


def test_window_march_leaves_the_trajectory_alone(dev, monkeypatch):
    """march_window = 'side' (the rest of a refresh window drawn and marched as one series of launches on a side stream, beside the
    refresh iteration's step), 'main' (the same series on the compute stream) and 'off' (every iteration marches in place) over 40
    iterations -- three grid refreshes, two batch-size updates: the same batches, the same RNG call indices per iteration,
    bit-identical parameters and sampler state; the window forms have marched the iterations up to the next refresh already."""
    from xrnerf_amd.train import Trainer
    out = []
    for mode in ('side', 'main', 'off'):
        tr = Trainer(dev, n_img=3, H=128, W=128, seed=5, march_window=mode, native_loop=False)
        assert tr.march_window == mode
        hist = []
        for _ in range(40):
            tr.step()
            hist.append((tr.net.sampler.n_rays_per_batch, tr.net.sampler.k1_calls, tr.data.batches_drawn))
        torch.cuda.synchronize()
        out.append(([p.detach().clone() for p in tr.net.parameters()], tr.net.sampler.density_grid.clone(),
                    tr.net.sampler.density_grid_bitfield.clone(), hist, tr.samples_done))
    ref = out[2]
    for o in out[:2]:
        assert [h[0] for h in o[3]] == [h[0] for h in ref[3]]              # rays per batch, iteration by iteration
        assert o[3][-1][1] == ref[3][-1][1] + 8 and o[3][-1][2] == ref[3][-1][2] + 8     # iterations 40..47 are marched already
        assert o[3][15][1] == ref[3][15][1] and o[3][31][1] == ref[3][31][1]            # nothing is marched across a refresh
        for a, b in zip(o[0], ref[0]):
            assert torch.equal(a, b)
        assert torch.equal(o[1], ref[1]) and torch.equal(o[2], ref[2]) and o[4] == ref[4]


def test_frame_rendered_inside_a_window_rewinds_the_marches(dev):
    """The reference's hidden K1 generator is shared by training and test launches: a frame rendered between iterations j - 1 and j
    moves the jitter of iteration j on.  With the window marched ahead, a test-mode launch takes the marches of j.. back (RNG call
    index, batch cursor) and the trainer marches what is left of the window again: bit-identical trajectory to marching in place,
    on the per-iteration path and through the native loop."""
    from xrnerf_amd.train import Trainer, render_frame
    out = []
    for mode, native in (('off', False), ('side', False), ('side', True), ('main', True)):
        tr = Trainer(dev, n_img=3, H=128, W=128, seed=9, march_window=mode, native_loop=native)
        frames = []
        for stop in (5, 16, 23, 31, 36):
            tr.run(stop - tr.iter)
            rgb, _ = render_frame(tr.net, tr.data.poses[1], 48, 48, tr.data.focal * 48 / 128)
            frames.append(rgb.clone())
        tr.run(40 - tr.iter)
        torch.cuda.synchronize()
        out.append(([p.detach().clone() for p in tr.net.parameters()], frames, tr.net.sampler.density_grid_bitfield.clone(),
                    tr.net.sampler.n_rays_per_batch, tr.rays_done, tr.samples_done))
    ref = out[0]
    for o in out[1:]:
        for a, b in zip(o[0], ref[0]):
            assert torch.equal(a, b)
        for a, b in zip(o[1], ref[1]):
            assert torch.equal(a, b)
        assert torch.equal(o[2], ref[2]) and o[3:] == ref[3:]


def _trainer_state(tr):
    table = tr.net.mlp.embedder_pos.params
    st = tr.opt.state[table]
    return ([p.detach().clone() for p in tr.net.parameters()] + [st['m'].clone(), st['v'].clone(), st['ema'].clone()],
            (st['step'], tr.iter, tr.net.sampler.k1_calls, tr.net.sampler.k6_calls, tr.data.batches_drawn, tr.data.cur_i,
             tr.net.sampler.n_rays_per_batch, tr.rays_done, tr.samples_done),
            tr.net.sampler.density_grid.clone(), tr.net.sampler.density_grid_bitfield.clone())


def test_native_loop_between_refreshes_equals_the_per_iteration_path(dev, monkeypatch):
    """xr_ngp_loop_run enqueues the iterations between two grid refreshes from native code (the steps on the marched window, with the
    updates inside).  41 iterations -- refreshes at 0, 16, 32, two batch-size updates -- as (a) per-iteration Python
    path, (b) Trainer.run over whole windows, (c) single step() calls that go through the native loop one iteration at a time,
    (d) windows cut at odd places with a multi-stage KernelTimer forcing the per-iteration path in between (marched batches are
    handed over in both directions): parameters, Adam moments, EMA copies, occupancy grids, every RNG / batch counter and the
    logged loss of the last iteration bit for bit."""
    from xrnerf_amd import ops
    from xrnerf_amd.train import Trainer
    out = []
    for mode in ('python', 'run', 'step', 'mixed'):
        monkeypatch.setenv('XRNERF_TRAINER', 'native_loop=%d' % (mode != 'python'))       # (through the environment override, once)
        tr = Trainer(dev, n_img=3, H=128, W=128, seed=7)
        assert tr.native_loop == (mode != 'python')
        if mode in ('python', 'step'):
            for _ in range(41):
                last = tr.step()
        elif mode == 'run':
            tr.run(7)
            tr.run(30)
            last = tr.run(4)
        else:
            tr.run(5)                                         # 0 (Python, marches 1..15) + 1..4 native
            ops.TIMER = ops.KernelTimer(only={'xr_hashgrid_fwd', 'xr_nerf_mlp_fwd'}, train_only=True)     # two stages: per-iteration path
            try:
                tr.run(3)                                     # 5, 6, 7 through Python (the same queue of marched iterations)
            finally:
                ops.TIMER = None
            tr.run(2)                                         # 8, 9 native again
            ops.TIMER = ops.KernelTimer(only={'xr_hashgrid_bwd'}, train_only=True)                         # one stage: stays native
            try:
                tr.run(20)
                torch.cuda.synchronize()
                n_timed, ms, _ = ops.TIMER.summary()['xr_hashgrid_bwd']
                assert n_timed == 20 and 0.0 < ms < 200.0
            finally:
                ops.TIMER = None
            for _ in range(10):
                tr.step()
            last = tr.step()
        if mode != 'python':
            assert tr._loop is not None
        torch.cuda.synchronize()
        out.append(_trainer_state(tr) + (float(last['loss']), float(last['log_vars']['psnr'])))
    ref = out[0]
    assert ref[1][1] == 41
    for o in out[1:]:
        assert o[1] == ref[1]
        for a, b in zip(o[0], ref[0]):
            assert torch.equal(a, b)
        assert torch.equal(o[2], ref[2]) and torch.equal(o[3], ref[3])
        assert o[4] == ref[4] and o[5] == ref[5]


def test_native_loop_iteration_events_bracket_every_iteration(dev):
    """Trainer.run(k, iter_events=...) records one timing event in front of every iteration and one behind the last, on both paths"""
    from xrnerf_amd import ops
    from xrnerf_amd.train import Trainer
    tr = Trainer(dev, n_img=3, H=128, W=128, seed=2)
    tr.run(14)
    ev = [ops._CEvent() for _ in range(6)]
    tr.run(5, iter_events=ev)                                # 14, 15 native, 16 per-iteration (refresh), 17, 18 native
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(5)]
    assert all(0.0 < m < 100.0 for m in ms)
    assert ms[2] > max(ms[0], ms[1])                          # the refresh iteration is the long one


def test_row_bands_of_a_frame_give_the_whole_frames_pixels(dev):
    """Image-space sharding of a validation frame (HashNerfNetwork._render_rows with several ranks): every rank marches a band of
    image rows, and the band's rays draw the jitter they have in the WHOLE frame's chunk series (sampler.frame_ray0 ->
    xr_rays_sampler's rng_ray0), so the all-gathered image is the one-GPU frame bit for bit -- bands that do not start on a chunk
    boundary included -- and the hidden generator's call counter ends where the whole frame leaves it."""
    from xrnerf_amd import ops
    from xrnerf_amd.dist import row_band
    from xrnerf_amd.train import Trainer
    tr = Trainer(dev, n_img=3, H=128, W=128, ema=False)
    tr.run(20)
    net, H, W = tr.net, 150, 160                                    # 24 000 rays = 6 chunks of 4096 (the last one partial)
    o, d = ops.gen_rays(tr.data.poses[1], H, W, tr.data.focal * H / 128, tr.data.focal * H / 128, 0.5 * W, 0.5 * H, device=dev)
    frame = {'rays_o': o, 'rays_d': d, 'img_ids': torch.zeros((H * W, 1), dtype=torch.int32, device=dev)}
    net.chunk = 4096
    net.sampler.rewind_marches()                                    # (a test-mode launch takes the marches issued ahead back first)
    k1 = net.sampler.k1_calls
    with torch.no_grad():
        whole = net.batchify_forward(dict(frame), is_test=True)
        whole = {k: v.clone() for k, v in whole.items()}
        assert net.sampler.k1_calls - k1 == 6
        for world in (2, 3):
            parts = []
            for rank in range(world):
                row0, nrows = row_band(H, rank, world)
                band = {k: v[row0 * W:(row0 + nrows) * W] for k, v in frame.items()}
                net.sampler.k1_calls = k1
                net.sampler.frame_ray0 = row0 * W
                try:
                    parts.append({k: v.clone() for k, v in net.batchify_forward(band, is_test=True).items()})
                finally:
                    net.sampler.frame_ray0 = 0
            for key in ('rgb', 'alpha'):
                assert torch.equal(torch.cat([p[key] for p in parts], 0), whole[key]), (world, key)
    assert float(whole['alpha'].max()) > 0.5

def test_backward_over_every_row_mode_through_the_native_loop(dev):
    """XR_MLP_LIVE=0 (read once per process: a child process here) runs the MLP backward and the scatter over every marched row -- same
    results as the live-row list -- through the native loop, at the speed of the list-less step (a start point of the march that depends
    on a kernel this mode does not launch once cost it 1.5 ms per iteration)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import json, torch, sys; sys.path.insert(0, %r)\n"
            "from xrnerf_amd.train import Trainer\n"
            "tr = Trainer(torch.device('cuda:0'), n_img=3, H=128, W=128, seed=7)\n"
            "tr.run(20); torch.cuda.synchronize()\n"
            "a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)\n"
            "a.record(); tr.run(3); tr.run(8); b.record(); torch.cuda.synchronize()\n"          # 20..30: no refresh inside
            "t = tr.net.mlp.embedder_pos.params\n"
            "print(json.dumps({'ms': a.elapsed_time(b) / 11, 'sum': float(t.double().sum()), 'abs': float(t.double().abs().sum())}))\n" % root)
    out = {}
    for live in ('1', '0'):
        env = dict(os.environ, XR_MLP_LIVE=live)
        r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        out[live] = json.loads(r.stdout.strip().splitlines()[-1])
    assert out['0']['ms'] < 1.0 and out['1']['ms'] < 1.0, out
    assert abs(out['0']['sum'] - out['1']['sum']) <= 1e-6 * out['1']['abs'], out
