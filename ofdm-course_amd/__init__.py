"""MI355X-native OFDM hot path (package directory; import it as `ofdm_course_amd`)."""
