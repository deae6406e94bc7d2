"""azdopt_amd -- MI355X-native self-play hot path of ariasanovsky/azdopt.

Host-side mirror of the reference's operator surface over the C ABI in include/azdopt_amd.h:
NablaOptimizer (optimizer/mod.rs), NablaModel / ActionModel / TrivialModel (model/), the c21
space ROTModifyParentsOnce and the ActionSet path with its symmetry axioms."""
from ._lib import ACT_NONE, ACT_RELU, ACT_SIGMOID, AzdError, build, lib  # noqa: F401
from .model import ActionModel, HashStreamModel, NablaModel, TrivialModel  # noqa: F401
from .optimizer import ArgminData, DenseArgminData, NablaOptimizer, RamseyArgminData, TreeView, tree_capacities  # noqa: F401
from . import sinks  # noqa: F401
from .space import Layered  # noqa: F401
from .space import DenseGraphSpace  # noqa: F401
from .space import ActionMultiset, ActionOrderIndependent, ActionSequence, ActionSet, OrderedActionSet, ActionsNeverRepeat, RamseySpaceNoEdgeRecolor, ROTModifyParentsOnce  # noqa: F401


def device_count():
    return lib().azd_device_count()
