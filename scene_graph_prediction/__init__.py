"""Import-path shim: `python -m scene_graph_prediction.main` resolves to the MI355X-native implementation."""
