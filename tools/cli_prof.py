import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle_amd.scene_graph_prediction import main as cli
os.chdir('/tmp')
pr = cProfile.Profile()
pr.enable()
cli.main(['--config', 'oracle_mv_learned.json', '--model_path', 'synthetic:full', '--limit_frames', '16'])
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45)
print(s.getvalue()[:9000])
