from .base import BasePrivacyAccountant, PrivacyAccountant, PrivacySpent
from .gaussian import GaussianAccountant
from .rdp import RDPAccountant, SubsampledRDPAccountant

__all__ = [
    "PrivacySpent",
    "PrivacyAccountant",
    "BasePrivacyAccountant",
    "GaussianAccountant",
    "RDPAccountant",
    "SubsampledRDPAccountant",
]
