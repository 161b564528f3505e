"""Host-side mirror of the reference's ``models`` package (lib/models/__init__.py:16-23): the factories
tools/test.py:87 reaches with ``eval('models.'+cfg.MODEL.NAME+'.get_pose_net')(cfg, is_train=False)``."""
from . import hrnet  # noqa: F401
from . import backbone  # noqa: F401
from . import transpose_h  # noqa: F401
from . import hrformer  # noqa: F401
from . import interformer_pureMulti  # noqa: F401
from . import interformer  # noqa: F401
from . import interformer_2stage  # noqa: F401
