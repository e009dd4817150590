"""gpyreg_amd -- MI355X-native dense GP core behind the gpyreg plugin API."""
