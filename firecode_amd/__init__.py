"""firecode_amd -- MI355X (gfx950) implementation of FIRECODE's ensemble-geometry
hot path behind the names FIRECODE imports.

Python here is host glue only: argument normalisation, logging hooks and the
reference's return conventions.  All arithmetic runs in hand-written HIP
kernels inside ``libfc_hip.so`` (C ABI in ``include/fc_hip.h``); there is no
NumPy fallback -- without the library or without a gfx950 device every compute
call raises ``FirecodeHipError``.
"""

from firecode_amd import _lib  # noqa: F401
from firecode_amd._lib import (  # noqa: F401
    DeviceEnsemble,
    FirecodeHipDeviceError,
    FirecodeHipError,
    FirecodeHipInputError,
    device_count,
    device_info,
    init,
    pinned_empty,
    shutdown,
)
from firecode_amd import (  # noqa: F401,E402
    algebra, embeds, ensemble, host_helpers, hypermolecule_class, operators, pruner, pt, refining, rmsd,
    torsion_module, utils,
)

__version__ = "0.1.0"
