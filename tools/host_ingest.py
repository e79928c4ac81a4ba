"""Host-ingest ceiling: how many frames per second can the HOST decode for the GPUs?
CPU only -- no GPU call, no model. The work timed is exactly what stays on the host in the drop-in CLI: per camera view
`OracleWrapper._open_rgb` (= the reference's `Image.open(path).convert('RGB')`, scene_graph_prediction_model_oracle.py:60-63)
plus `np.asarray` (the hand-over to the device preprocessor, which does pad + resize + normalise on the card), on the JPEG
files the synthetic dataset writes at the 4D-OR camera size (2048 x 1536, quality 90, ~700 KB).

P processes (one per GPU rank, as the multi-GPU CLI runs) x T decode threads each (the wrapper's pool) share the cores this
job may use; every process decodes `frames` frames of `views` views; reported: frames/s per process and summed.
usage: host_ingest.py [views=4] [frames=48] [threads=8] [procs=1,2,4,8] [WxH=2048x1536]"""
import json
import multiprocessing as mp
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _paths(n_images, size):
    from oracle_amd.scene_graph_prediction.scene_graph_helpers.dataset.synthetic_dataset import SyntheticORDataset
    ds = SyntheticORDataset.__new__(SyntheticORDataset)       # only the image writer is needed
    ds.root, ds.image_size, ds.image_format, ds.unique_images, ds._written = '/tmp/orc_host_ingest', size, 'jpg', 1 << 30, set()
    return [ds._image(i // 6, i % 6) for i in range(n_images)]


def _worker(rank, paths, views, frames, threads, barrier, out):
    from concurrent.futures import ThreadPoolExecutor

    import numpy as np

    from oracle_amd.scene_graph_prediction.scene_graph_helpers.model.scene_graph_prediction_model_oracle import OracleWrapper

    def one(p):
        return np.asarray(OracleWrapper._open_rgb(p)).shape

    pool = ThreadPoolExecutor(max_workers=threads) if threads > 0 else None
    todo = [paths[(rank * 7 + i) % len(paths)] for i in range(frames * views)]
    for p in todo[:views]:
        one(p)                                                 # page the files and libjpeg in
    barrier.wait()
    t0 = time.perf_counter()
    if pool is None:
        for p in todo:
            one(p)
    else:
        list(pool.map(one, todo))
    out.put((rank, time.perf_counter() - t0))


def main():
    views = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    threads = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    procs = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else '1,2,4,8').split(',')]
    w, h = (int(x) for x in (sys.argv[5] if len(sys.argv) > 5 else '2048x1536').split('x'))
    paths = _paths(24, (w, h))
    kb = sum(os.path.getsize(p) for p in paths) / len(paths) / 1024
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count()
    res = {'views': views, 'frames_per_process': frames, 'threads_per_process': threads, 'image': f'{w}x{h} JPEG q90, {kb:.0f} KB mean',
           'cpus_visible': os.cpu_count(), 'cpus_in_affinity': cores, 'OMP_NUM_THREADS': os.environ.get('OMP_NUM_THREADS'), 'curve': []}
    ctx = mp.get_context('spawn')
    for P in procs:
        barrier, out = ctx.Barrier(P), ctx.Queue()
        ps = [ctx.Process(target=_worker, args=(r, paths, views, frames, threads, barrier, out)) for r in range(P)]
        for p in ps:
            p.start()
        secs = [out.get()[1] for _ in ps]
        for p in ps:
            p.join()
        slow = max(secs)
        row = {'processes': P, 'seconds_slowest': slow, 'frames_per_s_per_process': frames / slow, 'frames_per_s_total': P * frames / slow,
               'ms_per_view_per_thread': 1e3 * slow * max(threads, 1) / (frames * views)}
        res['curve'].append(row)
        print(json.dumps(row), flush=True)
    print(json.dumps({'host_ingest': res}), flush=True)


if __name__ == '__main__':
    main()
