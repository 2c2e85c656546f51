"""Privacy error hierarchy (reference: nanofed/privacy/exceptions.py:1-22).

These deliberately do not derive from ``NanoFedError``: callers of the
reference catch ``PrivacyError`` separately.
"""


class PrivacyError(Exception):
    """Root of privacy-related failures."""


class PrivacyBudgetExceededError(PrivacyError):
    """The configured (epsilon, delta) budget has been used up."""


class PrivacyConfigurationError(PrivacyError):
    """A privacy configuration is inconsistent."""


class NoiseGenerationError(PrivacyError):
    """A noise generator failed while producing a sample."""
