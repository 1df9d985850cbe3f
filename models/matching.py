from linetr_amd.matching import Matching  # noqa: F401
