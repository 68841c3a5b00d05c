"""Channel: one reach between two boundaries, its cross-sections and its initial state
(reference: src/hydromodel/channel.py:7-390).

After `initialize_conditions(n_nodes)` the object holds
  ch_at_node[N]            node chainages (np.linspace between the boundaries)
  xs_at_node[N]            one TrapezoidalSection per node (interpolated between the input sections)
  node_geometry            the same as a dict of [N] arrays - the FS_GEO_* table the kernel reads
  initial_conditions[N,2]  depth and flow at t = 0 ('linear' | 'GVF_equation' | 'steady-state')
"""
import numpy as np

from . import hydraulics
from .boundary import Boundary
from .cross_section import CrossSection, TrapezoidalSection, interpolate_cross_section, section_table


class Channel:
    def __init__(self, upstream_boundary: Boundary, downstream_boundary: Boundary, initial_flow: float,
                 roughness: float = None, width: float = None, interpolation_method: str = 'GVF_equation'):
        if interpolation_method not in ('linear', 'GVF_equation', 'steady-state'):
            raise ValueError("Invalid interpolation method.")
        self.interpolation_method = interpolation_method
        self.initial_conditions = None
        self.conditions_initialized = False
        self.initial_flow_rate = initial_flow
        self.roughness, self.width = roughness, width
        self.upstream_boundary, self.downstream_boundary = upstream_boundary, downstream_boundary
        self.length = downstream_boundary.chainage - upstream_boundary.chainage
        self.xs_chainages = self.input_xs = None
        self.ch_at_node = self.xs_at_node = self.node_geometry = None
        self.coords_chainages = self.coords = None

    # ---- geometry input ------------------------------------------------------------------
    def set_coords(self, coords, chainages):
        """Centreline x,y pairs and their chainages; used for the curvature of the input sections."""
        self.coords_chainages = np.asarray(chainages, dtype=np.float64)
        self.coords = np.asarray(coords, dtype=np.float64)
        self.coordinated = True

    def set_cross_sections(self, chainages, sections):
        chainages = np.asarray(chainages, dtype=float)
        if len(chainages) != len(sections):
            raise ValueError("chainages and sections must have same length")
        if not np.all(np.diff(chainages) > 0):
            raise ValueError("chainages must be strictly increasing")
        self.xs_chainages, self.input_xs = chainages, sections

    # ---- per-node accessors (channel.py:172-190) --------------------------------------------
    def area_at(self, i, hw):
        return self.xs_at_node[i].area(hw)

    def hydraulic_radius(self, i, hw):
        return self.xs_at_node[i].hydraulic_radius(hw)

    def top_width(self, i, hw):
        return self.xs_at_node[i].top_width(hw)

    def bed_level_at(self, i):
        return self.xs_at_node[i].z_min

    def dA_dh(self, i, hw):
        return self.xs_at_node[i].dA_dh(hw=hw)

    def Se(self, h, Q, i):
        """energy slope at node i: friction + transverse circulation (channel.py:53-69)"""
        xs = self.xs_at_node[i]
        return xs.friction_slope(h=h, Q=Q) + xs.curvature_slope(h=h, Q=Q)

    def dSe_dA(self, h, Q, i):
        """channel.py:71-87.  Mixed convention, reproduced: the friction part is per unit area, the curvature part comes
        already multiplied by dA/dh (cross_section.py:164) - and the Jacobian multiplies the sum by dA/dh once more."""
        xs = self.xs_at_node[i]
        return xs.dSf_dA(h=h, Q=Q) + xs.dSc_dA(h=h, Q=Q)

    def dSe_dQ(self, h, Q, i):
        """channel.py:89-105"""
        xs = self.xs_at_node[i]
        return xs.dSf_dQ(h=h, Q=Q) + xs.dSc_dQ(h=h, Q=Q)

    # ---- set-up ------------------------------------------------------------------------------
    def initialize_conditions(self, n_nodes: int) -> None:
        self._initialize_geometry(n_nodes)
        self.initial_conditions = np.zeros((n_nodes, 2), dtype=np.float64)
        Q = self.initial_flow_rate
        {'linear': self._linear_conditions, 'GVF_equation': self._gvf_conditions,
         'steady-state': self._steady_conditions}[self.interpolation_method](n_nodes, Q)
        self.conditions_initialized = True

    def _initialize_geometry(self, n_nodes):
        if self.xs_chainages is None or self.input_xs is None:
            self._provisional_sections()
        self.ch_at_node = np.linspace(self.upstream_boundary.chainage, self.downstream_boundary.chainage, n_nodes)
        if self.coords_chainages is not None and self.coords is not None:
            self._input_curvatures()
        xc, xs_in = self.xs_chainages, self.input_xs
        nodes = []
        for s in self.ch_at_node:
            if s <= xc[0]:
                nodes.append(xs_in[0])
            elif s >= xc[-1]:
                nodes.append(xs_in[-1])
            else:
                j = int(np.searchsorted(xc, s)) - 1
                nodes.append(interpolate_cross_section(xs_in[j], xs_in[j + 1], dist1=s - xc[j], dist2=xc[j + 1] - s))
        self.xs_at_node = nodes
        self.node_geometry = section_table(nodes)
        self.upstream_boundary.cross_section = nodes[0]
        self.downstream_boundary.cross_section = nodes[-1]

    def _provisional_sections(self):
        """Two rectangles from width / roughness / boundary bed levels (channel.py:282-294)."""
        us, ds = self.upstream_boundary, self.downstream_boundary
        a = TrapezoidalSection(b_main=self.width, m_main=0, z_bed=us.bed_level, n_main=self.roughness)
        b = TrapezoidalSection(b_main=self.width, m_main=0, z_bed=ds.bed_level, n_main=self.roughness)
        a.bed_slope = b.bed_slope = (a.z_min - b.z_min) / self.length
        us.cross_section, ds.cross_section = a, b
        self.xs_chainages = [us.chainage, ds.chainage]
        self.input_xs = [a, b]

    def _input_curvatures(self):
        """Signed curvature at the interior input sections from the turning angle of the centreline
        between neighbouring sections (channel.py:243-277)."""
        xc = self.xs_chainages
        for i in range(1, len(self.input_xs) - 1):
            chs = np.array([xc[i - 1], xc[i], xc[i + 1]])
            pts = np.column_stack([np.interp(chs, self.coords_chainages, self.coords[:, k]) for k in (0, 1)])
            v1, v2 = pts[1] - pts[0], pts[2] - pts[1]
            l1, l2 = np.linalg.norm(v1), np.linalg.norm(v2)
            if l1 == 0 or l2 == 0:
                kappa = 0.0
            else:
                ang = np.arccos(np.clip(np.dot(v1, v2) / (l1 * l2), -1.0, 1.0))
                kappa = 2 * np.sin(ang / 2) / (0.5 * (l1 + l2)) * np.sign(v1[0] * v2[1] - v1[1] * v2[0])
            self.input_xs[i].curvature = kappa

    # ---- initial conditions ------------------------------------------------------------------------
    def _steady_conditions(self, n, Q):
        """Normal depth at every node (channel.py:296-305)."""
        for i, xs in enumerate(self.xs_at_node):
            if xs.bed_slope is None:
                raise ValueError("Bed slope must be defined.")
            self.initial_conditions[i] = (xs.normal_depth(Q_target=Q), Q)

    def _linear_conditions(self, n, Q):
        h0, hN = self.upstream_boundary.initial_depth, self.downstream_boundary.initial_depth
        for i in range(n):
            x = self.length * i / (n - 1)
            self.initial_conditions[i] = (h0 + (hN - h0) * x / self.length, Q)

    def _gvf_conditions(self, n, Q):
        """Backwater curve from the downstream depth by a predictor-corrector (Heun) march on
        dh/dx = (S0 - Se) / (1 - Fr^2)  (channel.py:307-378)."""
        dx = self.length / (n - 1)

        def slope(h_in, i, S0):
            hw = h_in + self.bed_level_at(i)
            A, T = self.area_at(i, hw), self.top_width(i, hw)
            if T < 1e-6 or A < 1e-6:
                return 0.0
            Fr = hydraulics.froude_num(T=T, A=A, Q=Q)
            if Fr > 1.0:
                raise RuntimeError(f"GVF Error: Flow became supercritical (Fr={Fr:.2f}) at node {i}. "
                                   "Downstream boundary control is not valid for this Q.")
            den = 1 - Fr ** 2
            if den < 0.01:
                print(f"Warning: GVF approaching critical depth at node {i} (Fr={Fr:.2f}). Clamping slope.")
                den = 0.01
            return (S0 - self.Se(h=h_in, Q=Q, i=i)) / den

        h = self.downstream_boundary.initial_depth
        self.initial_conditions[n - 1] = (h, Q)
        for i in reversed(range(n - 1)):
            # bed slope of the interval being crossed, used by predictor and corrector alike (channel.py:344)
            S0 = (self.bed_level_at(i) - self.bed_level_at(i + 1)) / dx
            k1 = slope(h, i + 1, S0)
            h_pred = h - k1 * dx
            if h_pred <= 0:
                h_pred = 0.01
            k2 = slope(h_pred, i, S0)
            h_new = h - 0.5 * (k1 + k2) * dx
            if h_new <= 0:
                print(f"Warning: GVF calculation resulted in h <= 0 at node {i}. Setting to 0.01.")
                h_new = 0.01
            h = h_new
            self.initial_conditions[i] = (h, Q)
