"""Host-side mirror of the reference's scene_graph_prediction package for the inference hot path."""
