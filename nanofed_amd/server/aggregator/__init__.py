from .base import AggregationResult, BaseAggregator
from .fedavg import FedAvgAggregator, weighted_state_sum
from .privacy import (
    PrivacyAwareAggregationConfig,
    PrivacyAwareAggregator,
    SecureAggregationType,
    ThresholdSecureAggregation,
)
from .secure import (
    BaseSecureAggregator,
    HomomorphicSecureAggregator,
    PairwiseMaskSecureAggregation,
    SecureAggregationConfig,
    SecureMaskingAggregator,
)

__all__ = [
    "AggregationResult",
    "BaseAggregator",
    "FedAvgAggregator",
    "weighted_state_sum",
    "PrivacyAwareAggregator",
    "PrivacyAwareAggregationConfig",
    "SecureAggregationType",
    "ThresholdSecureAggregation",
    "SecureAggregationConfig",
    "SecureMaskingAggregator",
    "BaseSecureAggregator",
    "HomomorphicSecureAggregator",
    "PairwiseMaskSecureAggregation",
]
