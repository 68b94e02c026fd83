"""Host side of the KiloNeRF path without a GPU: registry contract on the reference's finetune config, the packed
parameter block layout against the C-ABI's size, state-dict names, and loud failure without a device."""
import copy
import json
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
LAYERS = ['pts_linears.0', 'pts_linears.1', 'alpha_linear', 'feature_linear', 'direction_layer', 'rgb_linear']


def _state(gold):
    sd = {}
    for nm in LAYERS:
        sd[nm + '.weight'], sd[nm + '.bias'] = torch.tensor(gold['w.' + nm]), torch.tensor(gold['b.' + nm])
    return sd


def test_reference_config5_builds(tmp_path):
    import xrnerf_amd
    from xrnerf_amd import kilo, vanilla
    gold = np.load(os.path.join(G, 'ref_kilonerf.npz'))
    cfg = json.load(open(os.path.join(G, 'kilo_model_cfg.json')))
    p = '/root/reference/configs/kilonerf/kilonerf_finetune_Synthetic_NeRF_base01.py'
    if os.path.exists(p):
        import runpy
        ref = runpy.run_path(p)
        assert json.loads(json.dumps(ref['model'])) == cfg['model'] and ref['resolution_table']['Lego'] == kilo.LEGO_RESOLUTION
    model = copy.deepcopy(cfg['model'])
    torch.save(torch.tensor(gold['occupancy']), tmp_path / 'occupancy.pth')
    torch.save({'domain_mins': torch.tensor(gold['domain_mins']), 'domain_maxs': torch.tensor(gold['domain_maxs']), 'state_dict': _state(gold)},
               tmp_path / 'checkpoint.pth')
    model['mlp'].update(occupancy_checkpoint=str(tmp_path / 'occupancy.pth'), distilled_checkpoint=str(tmp_path / 'checkpoint.pth'),
                        resolution=[int(v) for v in gold['res']])
    net = xrnerf_amd.build_network(model)
    assert isinstance(net, kilo.KiloNerfNetwork) and isinstance(net.mlp, kilo.KiloNerfMLP)
    assert isinstance(net.render, vanilla.NerfRender) and net.N_importance == 0 and net.chunk == 40000
    assert net.mlp.embedder.get_embed_ch() == (63, 27)
    keys = set(net.state_dict().keys())
    assert {'mlp.multi_network.pts_linears.0.weight', 'mlp.multi_network.direction_layer.bias',
            'mlp.multi_network.rgb_linear.weight', 'mlp.multi_network.alpha_linear.bias'} <= keys
    assert len(net.mlp.get_view_dependent_parameters()) == 4
    # a checkpoint without the multi network's state is refused with an explanation, not mis-read
    torch.save({'root_nodes': []}, tmp_path / 'pickle.pth')
    model['mlp']['distilled_checkpoint'] = str(tmp_path / 'pickle.pth')
    with pytest.raises(NotImplementedError):
        xrnerf_amd.build_network(model)


def test_packed_block_layout():
    from xrnerf_amd import kilo, ops
    gold = np.load(os.path.join(G, 'ref_kilonerf.npz'))
    mn = kilo.MultiNetwork(24, 63, 27)
    mn.load_state_dict(_state(gold))
    p = mn.packed()
    assert p.shape == (24, ops.kilo_param_floats(10, 4, 2)) and p.shape[1] % 4 == 0
    n = 5
    off = 0
    assert np.array_equal(p[n, :63 * 32].numpy(), gold['w.pts_linears.0'][n].reshape(-1)); off += 63 * 32
    assert np.array_equal(p[n, off:off + 32].numpy(), gold['b.pts_linears.0'][n]); off += 32 + 32 * 32 + 32
    assert np.array_equal(p[n, off:off + 32].numpy(), gold['w.alpha_linear'][n, :, 0]) and float(p[n, off + 32]) == float(gold['b.alpha_linear'][n, 0])
    off += 36 + 32 * 32 + 32 + 59 * 32 + 32
    assert np.array_equal(p[n, off:off + 128].reshape(32, 4)[:, :3].numpy(), gold['w.rgb_linear'][n])
    assert np.array_equal(p[n, off + 128:off + 131].numpy(), gold['b.rgb_linear'][n]) and off + 132 == p.shape[1]
    # re-packed after an in-place parameter update
    with torch.no_grad():
        mn.rgb_linear.bias.add_(1.0)
    assert float(mn.packed()[n, off + 128]) == float(gold['b.rgb_linear'][n, 0]) + 1.0


def test_no_cpu_fallback_and_exports():
    from xrnerf_amd import _lib, kilo
    L = _lib.load()
    for name in ('xr_kilo_mlp_forward', 'xr_kilo_workspace_bytes', 'xr_kilo_param_floats', 'xr_nerf_render_forward'):
        assert hasattr(L, name)
    assert L.xr_kilo_workspace_bytes(1000, 24) >= 8000
    mlp, gmin, gmax = kilo.synthetic_scene('cpu', resolution=[48, 64, 32], gmin=[-1.2, -1.5, -0.8], gmax=[1.3, 1.4, 0.9])
    data = {'pts': torch.zeros(4, 8, 3), 'viewdirs': torch.ones(4, 3), 'global_domain_min': gmin, 'global_domain_max': gmax}
    with pytest.raises(_lib.XrError):
        mlp(data)


@pytest.mark.skipif(not os.path.isdir('/root/reference/xrnerf'), reason='needs the reference classes to write its checkpoint format')
def test_reference_pickled_distillation_checkpoint_is_read_without_the_reference_package(tmp_path):
    """a checkpoint written with the reference's OWN classes (Node tree with inner nodes, single-network MultiNetwork
    leaves, exactly what SaveDistillResultsHook stores) -> load_reference_distilled_checkpoint in a process state where
    those classes are not used for unpickling; merged weights = KiloNerfMLP.init_mlp's layout (kilonerf_mlp.py:46-127)"""
    import importlib
    import sys
    import types
    sys.path.insert(0, G)
    import ref_import
    ns = ref_import.load_kilo()
    if 'xrnerf.utils' not in sys.modules:
        m = types.ModuleType('xrnerf.utils'); m.__path__ = ['/root/reference/xrnerf/utils']; sys.modules['xrnerf.utils'] = m
    Node = importlib.import_module('xrnerf.utils.data_helper').Node
    torch.manual_seed(11)

    def leaf(lo, hi):
        n = Node(); n.domain_min, n.domain_max = lo, hi
        n.network = ns.MultiNetwork(1, 63, 27, 4, 32, 2, None, True, 32, 'relu', linear_implementation='bmm')
        return n
    a, b, c, d = leaf([0., 0, 0], [1., 1, 1]), leaf([1., 0, 0], [2., 1, 1]), leaf([0., 1, 0], [1., 2, 1]), leaf([1., 1, 0], [2., 2, 1])
    inner = Node(); inner.leq_child, inner.gt_child = c, d           # visited after the roots: order a, b, c, d
    path = str(tmp_path / 'distill.pth')
    torch.save({'root_nodes': [a, inner, b]}, path)
    from xrnerf_amd import kilo
    cp = kilo.load_reference_distilled_checkpoint(path)
    order = [a, b, c, d]
    assert cp['num_hidden_layers'] == 2
    assert np.array_equal(cp['domain_mins'].numpy(), np.float32([n.domain_min for n in order]))
    assert np.array_equal(cp['domain_maxs'].numpy(), np.float32([n.domain_max for n in order]))
    for i, n in enumerate(order):
        mods = dict(n.network.named_modules())
        for name in ('pts_linears.0', 'pts_linears.1', 'alpha_linear', 'feature_linear', 'direction_layer', 'rgb_linear'):
            assert torch.equal(cp['state_dict'][name + '.weight'][i], mods[name].weight.detach()[0].t())
            assert torch.equal(cp['state_dict'][name + '.bias'][i], mods[name].bias.detach()[0])
    # and through the registry class with the reference's constructor arguments
    torch.save(torch.ones(32 * 32 * 16, dtype=torch.bool), tmp_path / 'occupancy.pth')
    emb = dict(type='KiloNerfFourierEmbedder', num_networks=1, input_ch=3, multires=10, multires_dirs=4)
    with pytest.raises(NotImplementedError):
        kilo.KiloNerfMLP(resolution=[32, 32, 16], occupancy_checkpoint=str(tmp_path / 'occupancy.pth'), distilled_checkpoint=path, embedder=emb)
    mlp = kilo.KiloNerfMLP(resolution=[32, 32, 16], occupancy_checkpoint=str(tmp_path / 'occupancy.pth'), distilled_checkpoint=path,
                           embedder=emb, trust_pickle=True)
    assert mlp.multi_network.num_networks == 4 and mlp.multi_network.packed().shape == (4, 6248)
    assert torch.equal(mlp.multi_network.rgb_linear.weight[2], dict(c.network.named_modules())['rgb_linear'].weight.detach()[0].t())


def test_pack_and_unpack_round_trip_and_parameter_order():
    """the gradient blocks the backward kernel fills are cut back into per-parameter tensors by unpack_like: it must be the
    exact inverse of pack, in the order autograd hands the parameters over (= module registration order)"""
    from xrnerf_amd import kilo
    for nh, pc, dc in ((2, 63, 27), (1, 9, 3)):
        mn = kilo.MultiNetwork(5, pc, dc, num_hidden_layers=nh)
        ps = mn.ordered_parameters()
        assert [id(p) for p in ps] == [id(p) for p in mn.parameters()]
        blocks = kilo.MultiNetwork.pack([p.detach() for p in ps])
        assert torch.equal(blocks, mn.packed())
        back = kilo.MultiNetwork.unpack_like(blocks, ps)
        assert len(back) == len(ps) and all(a.shape == b.shape and torch.equal(a, b.detach()) for a, b in zip(back, ps))
        # padding slots (3 after b_alpha, the 4th rgb column, 1 after b_rgb) are zero in pack and ignored by unpack_like
        marked = blocks.clone() + 1.0
        back2 = kilo.MultiNetwork.unpack_like(marked, ps)
        assert sum(t.numel() for t in back2) == sum(p.numel() for p in ps) == blocks.numel() - 5 * (3 + 32 + 1)
