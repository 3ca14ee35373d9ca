import sys

import dig_amd.threedgraph as _impl
from dig_amd.threedgraph import dataset, evaluation, method, utils  # noqa: F401

sys.modules[__name__ + '.method'] = method
sys.modules[__name__ + '.utils'] = utils
sys.modules[__name__ + '.evaluation'] = evaluation
sys.modules[__name__ + '.dataset'] = dataset
