"""Child process of tests/test_parallel_gpu.py::test_rccl_single_rank_runs_the_sharded_branch.

Started fresh (no GPU call before the process group exists), it initialises torch.distributed with backend
"nccl" -- RCCL on ROCm -- for a world of ONE rank bound to cuda:0 and runs ShardedGallery's whole N > 1
branch with force_collectives=True: all_gather_into_tensor of the embeddings, dif_match into this rank's
packed record, the packed all-gather, dif_match_merge_packed.  Results go to an .npz the parent compares with
the plain Gallery.match and the reference-generated fixture.

    python rccl_single_rank_child.py <port> <out.npz>
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, 'deep-insight-face_amd'), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    port, out_path = sys.argv[1], sys.argv[2]
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = port
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import numpy as np
    import torch
    import torch.distributed as dist
    dev = torch.device('cuda', 0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)      # before any other GPU call
    try:
        torch.cuda.set_device(dev)
        import golden_inputs as gi
        from deep_insight_face import oneshot
        from deep_insight_face.parallel import ShardedGallery
        probes, gallery = gi.match_near_tie_inputs()
        res = {'backend': np.array(dist.get_backend()), 'world': np.array(dist.get_world_size()),
               'nccl_version': np.array(list(torch.cuda.nccl.version()))}
        sg = ShardedGallery(torch.from_numpy(gallery).to(dev), 0, force_collectives=True)
        plain = oneshot.Gallery(gallery)
        p_t = torch.from_numpy(probes).to(dev)
        for metric in (0, 1):
            for rep in range(2):                                   # the second pass reuses the step buffers
                idx, d = sg.match(p_t, metric)
            res['idx%d' % metric], res['d%d' % metric] = idx.cpu().numpy(), d.cpu().numpy()
            pi, pd = plain.match(probes, metric)
            res['plain_idx%d' % metric], res['plain_d%d' % metric] = pi, pd
        # the embeddings' all-gather by itself, and the allocation-free form
        g = sg.all_gather_embeddings(p_t)
        res['gathered_equal'] = np.array(bool(torch.equal(g, p_t)) and g.data_ptr() != p_t.data_ptr())
        j1, _ = sg.match(p_t, 1, copy=False)
        j2, _ = sg.match(p_t, 1, copy=False)
        res['copy_false_same_buffer'] = np.array(j1.data_ptr() == j2.data_ptr())
        # a ragged batch and a gallery that differs from the fixture's: the exchange is shape-agnostic
        i3, _ = sg.match(p_t[:5], 1)
        res['idx_ragged'] = i3.cpu().numpy()
        torch.cuda.synchronize()
        np.savez(out_path, **res)
    finally:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
