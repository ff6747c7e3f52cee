"""MI355X-native particle-filter update engine (drop-in for the MCL()/expected_pose() hot path of
AE-HYU/monte_carlo_localization).  See DESIGN.md."""
__version__ = "0.1.0"
