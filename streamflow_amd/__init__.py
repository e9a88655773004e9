"""streamflow_amd: MI355X-native StreamFlow hot path."""
