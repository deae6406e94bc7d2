"""Host-side description of the state/action spaces (the reference's trait surface).

`NablaStateActionSpace` (az-discrete-opt/src/nabla/space/mod.rs:5-39) methods are arbitrary Rust
in the reference; a device engine needs built-in device implementations, selected by id.  The
classes here carry the constants and the host-side closures (`init_states`), nothing more."""
import numpy as np

from . import _lib


class ActionsNeverRepeat:
    """Marker: az-discrete-opt/src/space/axioms.rs:7"""


class ActionOrderIndependent:
    """Marker: az-discrete-opt/src/space/axioms.rs:10"""


class ActionSet:
    """Path encoding keyed by the SET of actions taken (az-discrete-opt/src/path/set.rs:6-37).
    Licensed only for spaces that are ActionsNeverRepeat + ActionOrderIndependent
    (space/axioms.rs:16-19).  On the device it is a KW x u64 bit mask per node."""
    PATH_KIND = _lib.PATH_SET

    @staticmethod
    def licensed_for(space):
        return isinstance(space, ActionsNeverRepeat) and isinstance(space, ActionOrderIndependent)


class ActionMultiset:
    """Path encoding keyed by the multiset of actions (path/multiset.rs:5-42), licensed for
    ActionOrderIndependent spaces (axioms.rs:14).  On a space that is also ActionsNeverRepeat every
    multiplicity is 1, so identity, order and length coincide with ActionSet's: same device keys."""
    PATH_KIND = _lib.PATH_SET

    @staticmethod
    def licensed_for(space):
        return isinstance(space, ActionOrderIndependent) and isinstance(space, ActionsNeverRepeat)


class ActionSequence:
    """Path encoding keyed by the actions in the order taken (path/sequence.rs:3-36), licensed for
    every space.  Distinct orders are distinct nodes: no transpositions, the search graph is a tree."""
    PATH_KIND = _lib.PATH_SEQUENCE

    @staticmethod
    def licensed_for(space):
        return True


class OrderedActionSet(ActionSequence):
    """path/ord_set.rs:3-34: despite the name a Vec that push_unchecked appends to, i.e. the same
    key as ActionSequence; licensed for ActionsNeverRepeat spaces (axioms.rs:12)."""

    @staticmethod
    def licensed_for(space):
        return isinstance(space, ActionsNeverRepeat)


class ROTModifyParentsOnce(ActionsNeverRepeat, ActionOrderIndependent):
    """c21 space: rooted ordered trees on N vertices, each parent may be modified once
    (graph-state/src/rooted_tree/space.rs:14-125; axioms asserted at :124-125).
    cost = Conjecture2Dot1Cost {lambda_1, matching}; evaluate = squish(mu + lambda_1)
    (graph-state/examples/04-c21-tree.rs:58-74,96-105)."""

    SPACE_ID = _lib.SPACE_C21

    def __init__(self, n):
        self.n = int(n)
        L = _lib.lib()
        self.STATE_DIM = L.azd_c21_state_dim(self.n)    # space.rs:46
        self.ACTION_DIM = L.azd_c21_action_dim(self.n)  # space.rs:48
        self.KEY_WORDS = L.azd_c21_key_words(self.n)

    @property
    def ROOT_BYTES(self):
        return self.n

    def default_permitted_range(self):
        """04-c21-tree.rs:85: 5..=(ACTION / 2), clamped for tiny N"""
        hi = max(1, self.ACTION_DIM // 2)
        return min(5, hi), hi

    def generate_roots(self, seed, count, first_agent=0, epoch=0, kmin=None, kmax=None):
        """`init_states` of the driver (04-c21-tree.rs:108-112) with a seeded generator.
        Returns packed roots: parents u8 [count, n], permitted u64 [count, KEY_WORDS]."""
        lo, hi = self.default_permitted_range()
        kmin = lo if kmin is None else kmin
        kmax = hi if kmax is None else kmax
        parents = np.zeros((count, self.n), np.uint8)
        permitted = np.zeros((count, self.KEY_WORDS), np.uint64)
        _lib.check(_lib.lib().azd_c21_generate_roots(seed, epoch, first_agent, count, self.n, kmin, kmax,
                                                     _lib.ptr(parents), _lib.ptr(permitted)), "azd_c21_generate_roots")
        return parents, permitted

    @staticmethod
    def action(index):
        """(parent, child) of action `index` (ordered_edge.rs:40-42 via edge.rs:55-65)"""
        c = 2
        while c * (c + 1) // 2 - 1 <= index:
            c += 1
        return index - (c * (c - 1) // 2 - 1), c


class RamseySpaceNoEdgeRecolor(ActionsNeverRepeat, ActionOrderIndependent):
    """Ramsey space: C-colourings of the edges of K_N, an action recolours one still-permitted edge
    (graph-state/src/ramsey_counts/space.rs:10-176; axioms asserted at :179-186).
    cost = TotalCounts (monochromatic `sizes[c]`-cliques per colour); evaluate = sum_c count_c * w_c;
    g = c_s h + r (1 - h), h_sa = 1 - c*/c (:159-177).  Drivers: 01-r333.rs (N 16, [3,3,3]),
    02-r44.rs (N 17, [4,4])."""

    SPACE_ID = _lib.SPACE_RAMSEY

    def __init__(self, n, sizes, weights=None):
        self.n = int(n)
        self.sizes = [int(x) for x in sizes]
        self.weights = [1.0] * len(self.sizes) if weights is None else [float(x) for x in weights]
        self.C = len(self.sizes)
        self.E = self.n * (self.n - 1) // 2
        L = _lib.lib()
        self.STATE_DIM = L.azd_ramsey_state_dim(self.n, self.C)    # space.rs:40
        self.ACTION_DIM = L.azd_ramsey_action_dim(self.n, self.C)  # space.rs:42
        self.KEY_WORDS = L.azd_ramsey_key_words(self.n, self.C)

    @property
    def ROOT_BYTES(self):
        return self.E

    def default_permitted_range(self):
        """02-r44.rs:83: 12..=(E / 2), clamped to what one node can hold"""
        hi = max(1, min(self.E // 2, 128 // (self.C - 1)))
        return min(12, hi), hi

    def generate_roots(self, seed, count, first_agent=0, epoch=0, kmin=None, kmax=None):
        """`init_state` of the drivers with a seeded generator.  Returns packed roots:
        colors u8 [count, E], permitted edges u64 [count, KEY_WORDS]."""
        lo, hi = self.default_permitted_range()
        kmin = lo if kmin is None else kmin
        kmax = hi if kmax is None else kmax
        colors = np.zeros((count, self.E), np.uint8)
        permitted = np.zeros((count, self.KEY_WORDS), np.uint64)
        _lib.check(_lib.lib().azd_ramsey_generate_roots(seed, epoch, first_agent, count, self.n, self.C, kmin, kmax,
                                                        _lib.ptr(colors), _lib.ptr(permitted)), "azd_ramsey_generate_roots")
        return colors, permitted

    def action(self, index):
        """(edge position, new colour) of action `index` (space.rs:48-54)"""
        return index % self.E, index // self.E


class DenseGraphSpace(ActionsNeverRepeat, ActionOrderIndependent):
    """Connected graphs on N <= 64 vertices; an action adds an absent edge or deletes a present non-cut edge, every edge
    slot at most once (BASELINE configs[4], N = 50).  BUILD-DEFINED: the reference has the pieces
    (connected_bitset_graph/mod.rs: is_cut_edge, action_kinds, conjecture_2_1_cost; bitset_graph/space/action.rs:
    AddOrDeleteEdge indexing; 05-ah.rs:39-40: STATE = E + ACTION + 1) but no live NablaStateActionSpace over general
    graphs; the definition is in oracle/dense_graph.inc.  cost = Conjecture2Dot1Cost {lambda_1, matching number};
    evaluate = squish(mu + lambda_1) with the bounds of 04-c21-tree.rs:58-74.  `max_slots`: the most modifiable slots a root
    may bring (= legal actions a node can hold; the engine sizes its keys from it: 128 / 256 / 640 / 1024); the drivers'
    image is E // 2 (04-c21-tree.rs:85 permits up to half of ACTION_DIM), 128 keeps the keys at two words."""

    SPACE_ID = _lib.SPACE_DENSE

    def __init__(self, n, p=0.2, max_slots=128):
        self.n, self.p = int(n), float(p)
        self.MAX_SLOTS = int(max_slots)
        self.E = self.n * (self.n - 1) // 2
        L = _lib.lib()
        self.STATE_DIM = L.azd_dense_state_dim(self.n)
        self.ACTION_DIM = L.azd_dense_action_dim(self.n)
        self.KEY_WORDS = L.azd_dense_key_words(self.n)

    @property
    def ROOT_BYTES(self):
        return 8 * self.n

    def default_permitted_range(self):
        hi = min(self.MAX_SLOTS, self.E // 2)
        return min(5, hi), hi

    def generate_roots(self, seed, count, first_agent=0, epoch=0, kmin=None, kmax=None):
        """seeded `init_states`: (adj u64 [count, n] viewed as bytes [count, 8 n], modifiable slots u64 [count, KEY_WORDS])"""
        lo, hi = self.default_permitted_range()
        kmin = lo if kmin is None else kmin
        kmax = hi if kmax is None else kmax
        adj = np.zeros((count, self.n), np.uint64)
        slots = np.zeros((count, self.KEY_WORDS), np.uint64)
        _lib.check(_lib.lib().azd_dense_generate_roots(seed, epoch, first_agent, count, self.n, kmin, kmax, self.p,
                                                       _lib.ptr(adj), _lib.ptr(slots)), "azd_dense_generate_roots")
        return adj.view(np.uint8).reshape(count, 8 * self.n), slots

    def action(self, index):
        """("add" | "delete", edge slot) of action `index` (AddOrDeleteEdge::from_action_index, action.rs:20-27)"""
        return ("add", index) if index < self.E else ("delete", index - self.E)


class Layered:
    """Layered<LAYERS, Space> (az-discrete-opt/src/space/layered.rs:3-20; trait impl nabla/space/mod.rs:41-111):
    the evaluator sees the last `layers` states of the path.  STATE_DIM = layers * inner STATE_DIM; every other
    method is the inner space's on the newest state; the axioms are inherited (layered.rs:12-19)."""

    def __init__(self, space, layers):
        if not 2 <= int(layers) <= 8:
            raise ValueError("layers must be 2..8")
        self.space, self.layers = space, int(layers)
        self.STATE_DIM = space.STATE_DIM * self.layers  # nabla/space/mod.rs:53
        # marker axioms follow the inner space
        self.__class__ = type("Layered", (Layered,) + tuple(b for b in (ActionsNeverRepeat, ActionOrderIndependent)
                                                            if isinstance(space, b)), {})

    def __getattr__(self, name):  # ACTION_DIM, KEY_WORDS, n, SPACE_ID, generate_roots, ...
        return getattr(self.space, name)
