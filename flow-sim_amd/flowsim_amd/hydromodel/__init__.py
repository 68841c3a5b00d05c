"""Host-side mirror of the reference's public interface (src/hydromodel/*): same class names,
constructor arguments, attributes and error behaviour for the objects a case script touches on the
way to PreissmannSolver.run().  Everything here is setup / bookkeeping in numpy; the Newton loop
itself runs in the HIP kernels behind the C ABI (flowsim_amd._abi)."""
from .boundary import Boundary
from .channel import Channel
from .cross_section import CrossSection, IrregularSection, TrapezoidalSection, interpolate_cross_section
from .hydrograph import Hydrograph
from .lumped_storage import LumpedStorage
from .preissmann import PreissmannSolver
from .rating_curve import RatingCurve

__all__ = ["Boundary", "Channel", "CrossSection", "IrregularSection", "TrapezoidalSection", "interpolate_cross_section",
           "Hydrograph", "LumpedStorage", "PreissmannSolver", "RatingCurve"]
