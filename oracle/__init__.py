"""TEST INFRASTRUCTURE — CPU oracle for the ConAN-FGW hot path.  Never imported by the product package."""
