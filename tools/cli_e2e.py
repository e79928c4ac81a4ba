"""GPU box: the drop-in CLI end to end at full size on the synthetic dataset (host image decode, device preprocessing,
tokenisation, generate(), answer parsing, metrics): wall-clock frames/s including everything the bench leaves out."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle_amd.scene_graph_prediction import main as cli
from oracle_amd.scene_graph_prediction.scene_graph_helpers.model import scene_graph_prediction_model_oracle as wmod

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
model = sys.argv[2] if len(sys.argv) > 2 else 'synthetic:full'
os.chdir('/tmp')
t = {'fwd': 0.0, 'n': 0}
orig = wmod.OracleWrapper.forward
def timed(self, batch):
    s = time.perf_counter(); out = orig(self, batch); t['fwd'] += time.perf_counter() - s; t['n'] += len(batch); return out
wmod.OracleWrapper.forward = timed
s0 = time.perf_counter()
res = cli.main(['--config', 'oracle_mv_learned.json', '--model_path', model, '--limit_frames', str(n)])
wall = time.perf_counter() - s0
print(f'frames {t["n"]}  forward() total {t["fwd"]:.2f}s = {t["n"] / t["fwd"]:.2f} frames/s  (whole CLI incl. model build {wall:.1f}s)')
