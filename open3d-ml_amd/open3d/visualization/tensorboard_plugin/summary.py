"""``from open3d.visualization.tensorboard_plugin import summary`` (``semantic_segmentation.py:13``,
``object_detection.py:17``) — the pipelines import the module at load time and only call into it when a training run
asks for 3-D summaries.  Inference never does; the calls are inert here."""


def add_3d(*args, **kwargs):
    return None


def to_dict_batch(*args, **kwargs):
    return {}
