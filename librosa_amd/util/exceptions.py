"""Exception classes of the drop-in (same names and hierarchy as ``librosa/util/exceptions.py:6-15``)."""


class LibrosaError(Exception):
    """Root of the library's exception hierarchy."""


class ParameterError(LibrosaError):
    """Raised for mal-formed or out-of-range arguments."""
