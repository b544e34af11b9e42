"""MI355X-native text+vision fusion train step of Ultrafnd."""
