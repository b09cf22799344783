"""Import-compatible alias package: the reference's `clickstream_transformer.*` module paths resolve to
the MI355X-native implementation in `bert4clickpath_amd.clickstream_transformer`."""
