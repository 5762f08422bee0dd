"""Dataset inference + evaluation harness: the reference's benchmark protocol
(script/{depth,normals,iid}/{infer,eval}.py, src/dataset/*, src/util/{alignment,metric}.py) around the
MI355X engine.  ``datasets`` reads the benchmark layouts (folder or tar), ``harness`` holds the two
command-line programs, ``metrics`` / ``alignment`` the scores."""
from .alignment import align_depth_least_square, depth2disparity, disparity2depth  # noqa: F401
from .datasets import DatasetMode, PredNameMode, get_dataset, get_pred_name, load_dataset_config  # noqa: F401
from .metrics import MetricTracker  # noqa: F401
