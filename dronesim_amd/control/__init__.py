from .indi_fleet import BaseControl, INDIControl  # noqa: F401
