"""Host-side pieces around the MI355X surfel rasterizer: synthetic scenes, cameras, the
render() wrapper, the sparse-control-point deformation (PyTorch-ROCm) and the data-parallel
train step.  The operator itself lives in ``diff_surfel_rasterization`` (same import name as
the reference's extension package)."""
